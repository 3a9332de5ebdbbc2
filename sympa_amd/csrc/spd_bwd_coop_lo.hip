// spd backward, sixteen lanes per pair, M = 3..11 (spd_coop_bwd_kernel.hpp)
#include "spd_coop_bwd_kernel.hpp"

namespace sympa_hip {
void launch_spd_coop_bwd_lo(const SpdBwdArgs& a, int n, dim3 grid, hipStream_t s) {
    switch (n) {
        case 3: hipLaunchKernelGGL(spd_coop_bwd_kernel<3>, spd_coop_bwd_grid(a.b, 3), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<3>())); break;
        case 4: hipLaunchKernelGGL(spd_coop_bwd_kernel<4>, spd_coop_bwd_grid(a.b, 4), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<4>())); break;
        case 5: hipLaunchKernelGGL(spd_coop_bwd_kernel<5>, spd_coop_bwd_grid(a.b, 5), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<5>())); break;
        case 6: hipLaunchKernelGGL(spd_coop_bwd_kernel<6>, spd_coop_bwd_grid(a.b, 6), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<6>())); break;
        case 7: hipLaunchKernelGGL(spd_coop_bwd_kernel<7>, spd_coop_bwd_grid(a.b, 7), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<7>())); break;
        case 8: hipLaunchKernelGGL(spd_coop_bwd_kernel<8>, spd_coop_bwd_grid(a.b, 8), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<8>())); break;
        case 9: hipLaunchKernelGGL(spd_coop_bwd_kernel<9>, spd_coop_bwd_grid(a.b, 9), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<9>())); break;
        case 10: hipLaunchKernelGGL(spd_coop_bwd_kernel<10>, spd_coop_bwd_grid(a.b, 10), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<10>())); break;
        default: hipLaunchKernelGGL(spd_coop_bwd_kernel<11>, spd_coop_bwd_grid(a.b, 11), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<11>())); break;
    }
}
}  // namespace sympa_hip
