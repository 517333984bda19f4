// compile-only: resource usage of the lean solver stage (hipcc -c --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)
#include "../../robopianist_amd/csrc/rp_solver2.hpp"
template __global__ void rp_lean_solver_kernel<double>(RpModel<double>, RpState<double>, RpStage<double>);
