// upper model: egrad2rgrad / RSGD step with EIGHT lanes per table row (two rows per DPP row: SYMPA_COOP_HALF), M = 7, 8
#define SYMPA_COOP_HALF
#include "siegel_table_kernel.hpp"
#include "siegel_coop_table.hpp"

namespace sympa_hip {
namespace {
template <int OP>
int launch_op(int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps, const double* clip,
              double max_norm, int* outside, hipStream_t s) {
    constexpr int PPR = spd_coop::GROUPS_PER_WAVE;
    const int rounds = spd_coop::coop_rounds(b, 1, PPR);
    const dim3 grid((unsigned)((b + PPR * rounds - 1) / (PPR * rounds)));
    switch (n) {
        case 7: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 7, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 8: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 8, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "eight-lanes table operations cover dims 7, 8");
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}
}  // namespace

int launch_table_half_upper(int op, int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps,
                            const double* clip, double max_norm, int* outside, hipStream_t s) {
    switch (op) {
        case spd_coop::OP_PROJX: return launch_op<spd_coop::OP_PROJX>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
        case spd_coop::OP_RSGD: return launch_op<spd_coop::OP_RSGD>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
        case spd_coop::OP_SQNORM: return launch_op<spd_coop::OP_SQNORM>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
        default: return launch_op<spd_coop::OP_EGRAD2RGRAD>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
    }
}
}  // namespace sympa_hip
