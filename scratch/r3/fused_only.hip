// compile-only: the fused-substeps kernel (capsule build) and its clean-up kernel
#include "../../robopianist_amd/csrc/rp_kernels.hpp"
template __global__ void rp_fused_steps_kernel<double, 0>(RpModel<double>, RpState<double>, RpStage<double>, int);
template __global__ void rp_cleanup_steps_kernel<double, 0>(RpModel<double>, RpState<double>, RpStage<double>, int);
