// spd backward, sixteen lanes per pair, M = 3..11 (spd_coop_bwd_kernel.hpp)
#include "spd_coop_bwd_kernel.hpp"

namespace sympa_hip {
void launch_spd_coop_bwd_lo(const SpdBwdArgs& a, int n, dim3 grid, hipStream_t s) {
    switch (n) {
        case 3: hipLaunchKernelGGL(spd_coop_bwd_kernel<3>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        case 4: hipLaunchKernelGGL(spd_coop_bwd_kernel<4>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        case 5: hipLaunchKernelGGL(spd_coop_bwd_kernel<5>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        case 6: hipLaunchKernelGGL(spd_coop_bwd_kernel<6>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        case 7: hipLaunchKernelGGL(spd_coop_bwd_kernel<7>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        case 8: hipLaunchKernelGGL(spd_coop_bwd_kernel<8>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        case 9: hipLaunchKernelGGL(spd_coop_bwd_kernel<9>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        case 10: hipLaunchKernelGGL(spd_coop_bwd_kernel<10>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
        default: hipLaunchKernelGGL(spd_coop_bwd_kernel<11>, spd_coop_bwd_grid(a.b), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b)); break;
    }
}
}  // namespace sympa_hip
