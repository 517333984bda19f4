// compile-only: the full-capacity solver stage (trunk4 build)
#include "../../robopianist_amd/csrc/rp_kernels.hpp"
template __global__ void rp_stage_kernel<double, 1, 4>(RpModel<double>, RpState<double>, RpStage<double>, int, int);
