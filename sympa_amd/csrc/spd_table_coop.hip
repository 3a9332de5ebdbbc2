// spd egrad2rgrad / RSGD step with sixteen lanes per table row, M = 3..16 (spd_coop_table.hpp)
#include "spd_coop_bwd_kernel.hpp"
#include "spd_coop_table.hpp"

namespace sympa_hip {
namespace {
template <int OP>
void launch_op(int n, double* x, const double* g, double* out, int64_t b, double lr, double wd, const double* clip,
               double max_norm, int32_t* status, hipStream_t s) {
    const int rounds = spd_coop::coop_rounds(b);
    const dim3 grid((unsigned)((b + 4 * rounds - 1) / (4 * rounds)));
    switch (n) {
        case 3: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<3, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 4: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<4, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 5: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<5, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 6: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<6, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 7: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<7, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 8: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<8, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 9: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<9, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 10: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<10, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 11: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<11, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 12: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<12, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 13: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<13, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 14: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<14, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 15: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<15, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        case 16: hipLaunchKernelGGL((spd_coop::spd_coop_table_kernel<16, OP>), grid, dim3(64), 0, s, x, g, out, b, lr, wd, clip, max_norm, status, rounds); break;
        default: break;
    }
}
}  // namespace

void launch_spd_coop_table(int op, int n, double* x, const double* g, double* out, int64_t b, double lr, double wd,
                           const double* clip, double max_norm, int32_t* status, hipStream_t s) {
    if (op == spd_coop::OP_RSGD) launch_op<spd_coop::OP_RSGD>(n, x, g, out, b, lr, wd, clip, max_norm, status, s);
    else launch_op<spd_coop::OP_EGRAD2RGRAD>(n, x, g, out, b, lr, wd, clip, max_norm, status, s);
}
}  // namespace sympa_hip
