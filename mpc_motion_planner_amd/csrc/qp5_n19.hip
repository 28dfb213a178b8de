// qp5_n19.hip — the kernel pair of the reference-as-shipped discretisation (N = 19, one arm): k_qp3f<6, 1, 5> + k_qp5<6>, as a translation unit of
// their own (like qp3_n25.hip: the machine scheduler's strategy is an option of the whole compilation).  The Makefile builds this file with the strategy
// named there; mpcmp.hip (compiled with -DMPCMP_SPLIT_N19) only declares the two kernels.  A single-file build of mpcmp.hip alone (tools/) still
// contains everything.
#include <hip/hip_runtime.h>
#include "../../include/mpcmp.h"
#define MPCMP_V3_TU
#include "qp_kernel_v3.hpp"
#include "qp_kernel_v5.hpp"

namespace mpcmp {
template __global__ void k_qp3f<6, 1, 5>(mpcmp_config, WS, const Qp3Pat *, Xch, int, double *);
template __global__ void k_qp5<6>(mpcmp_config, WS, const Qp3Pat *, Xch, int, const double *);
}  // namespace mpcmp
