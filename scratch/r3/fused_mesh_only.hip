// compile-only: the fused-substeps kernel, hull build
#include "../../robopianist_amd/csrc/rp_kernels.hpp"
template __global__ void rp_fused_steps_kernel<double, 1>(RpModel<double>, RpState<double>, RpStage<double>, int);
