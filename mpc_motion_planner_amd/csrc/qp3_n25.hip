// qp3_n25.hip — the N = 25 instantiations of k_qp3f / k_qp3 (single arm and dual arm) as a translation unit of their own.
// The machine scheduler's strategy is an option of the whole compilation, and these register-bound kernels want another one than k_qp2 / k_qp5: the
// Makefile builds this file with -mllvm -amdgpu-sched-strategy=iterative-minreg (round 5: k_qp3<8, 2> 96 B of scratch instead of 144, -5 % per QP; round 3
// used max-ilp, +4 % over the default; either costs k_qp2 8 - 12 %) and mpcmp.hip (compiled with -DMPCMP_SPLIT_N25: extern template declarations of the
// four kernels) without.  A single-file build of mpcmp.hip
// alone (tools/stamps3.py, tools/ablate.py) still contains everything.
#include <hip/hip_runtime.h>
#include "../../include/mpcmp.h"
#define MPCMP_V3_TU
#include "qp_kernel_v3.hpp"

namespace mpcmp {
template __global__ void k_qp3f<8, 1>(mpcmp_config, WS, const Qp3Pat *, Xch, int, double *);
template __global__ void k_qp3f<8, 2>(mpcmp_config, WS, const Qp3Pat *, Xch, int, double *);
template __global__ void k_qp3<8, 1>(mpcmp_config, WS, const Qp3Pat *, Xch, int, const double *);
template __global__ void k_qp3<8, 2>(mpcmp_config, WS, const Qp3Pat *, Xch, int, const double *);
}  // namespace mpcmp
