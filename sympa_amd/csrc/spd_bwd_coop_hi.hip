// spd backward, sixteen lanes per pair, M = 12..15 (spd_coop_bwd_kernel.hpp)
#include "spd_coop_bwd_kernel.hpp"

namespace sympa_hip {
void launch_spd_coop_bwd_hi(const SpdBwdArgs& a, int n, dim3 grid, hipStream_t s) {
    switch (n) {
        case 12: hipLaunchKernelGGL(spd_coop_bwd_kernel<12>, spd_coop_bwd_grid(a.b, 12), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<12>())); break;
        case 13: hipLaunchKernelGGL(spd_coop_bwd_kernel<13>, spd_coop_bwd_grid(a.b, 13), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<13>())); break;
        case 14: hipLaunchKernelGGL(spd_coop_bwd_kernel<14>, spd_coop_bwd_grid(a.b, 14), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<14>())); break;
        default: hipLaunchKernelGGL(spd_coop_bwd_kernel<15>, spd_coop_bwd_grid(a.b, 15), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<15>())); break;
    }
}
}  // namespace sympa_hip
