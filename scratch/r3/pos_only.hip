// compile-only: the position stage (capsule build)
#include "../../robopianist_amd/csrc/rp_kernels.hpp"
template __global__ void rp_stage_kernel<double, 0, 0, RPK_MAXD, 0>(RpModel<double>, RpState<double>, RpStage<double>, int, int);
