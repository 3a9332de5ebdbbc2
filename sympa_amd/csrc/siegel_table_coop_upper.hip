// upper model: egrad2rgrad / RSGD step with sixteen lanes per table row, M = 9..16 (siegel_coop_table.hpp)
#include "siegel_table_kernel.hpp"
#include "siegel_coop_table.hpp"

namespace sympa_hip {
namespace {
template <int OP>
int launch_op(int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps, const double* clip,
              double max_norm, int* outside, hipStream_t s) {
    const int rounds = spd_coop::coop_rounds(b);
    const dim3 grid((unsigned)((b + 4 * rounds - 1) / (4 * rounds)));
    switch (n) {
        case 9: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 9, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 10: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 10, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 11: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 11, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 12: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 12, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 13: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 13, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 14: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 14, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 15: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 15, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        case 16: hipLaunchKernelGGL((siegel_coop::siegel_coop_table_kernel<sympa::MODEL_UPPER, 16, OP>), grid, dim3(64), 0, s, z, g, out, b, lr, wd, eps, clip, max_norm, outside, rounds); break;
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "sixteen-lanes table operations cover dims 9..16");
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}
}  // namespace

int launch_table_coop_upper(int op, int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps,
                            const double* clip, double max_norm, int* outside, hipStream_t s) {
    switch (op) {
        case spd_coop::OP_PROJX: return launch_op<spd_coop::OP_PROJX>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
        case spd_coop::OP_RSGD: return launch_op<spd_coop::OP_RSGD>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
        case spd_coop::OP_SQNORM: return launch_op<spd_coop::OP_SQNORM>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
        default: return launch_op<spd_coop::OP_EGRAD2RGRAD>(n, z, g, out, b, lr, wd, eps, clip, max_norm, outside, s);
    }
}
}  // namespace sympa_hip
