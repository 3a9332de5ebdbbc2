// spd backward, sixteen lanes per pair with the QL of two rounds run together (spd_coop_bwd2_kernel): M = 3..10
#include "spd_coop_bwd_kernel.hpp"

namespace sympa_hip {
bool launch_spd_bwd_coop2_lo(const SpdBwdArgs& a, int n, hipStream_t s) {
    const int rounds = spd_coop::coop_rounds(a.b, 1, 8);
    const dim3 grid((unsigned)((a.b + 8 * rounds - 1) / (8 * rounds)));
    switch (n) {
        case 3: hipLaunchKernelGGL(spd_coop_bwd2_kernel<3>, grid, dim3(64), 0, s, a, rounds); return true;
        case 4: hipLaunchKernelGGL(spd_coop_bwd2_kernel<4>, grid, dim3(64), 0, s, a, rounds); return true;
        case 5: hipLaunchKernelGGL(spd_coop_bwd2_kernel<5>, grid, dim3(64), 0, s, a, rounds); return true;
        case 6: hipLaunchKernelGGL(spd_coop_bwd2_kernel<6>, grid, dim3(64), 0, s, a, rounds); return true;
        case 7: hipLaunchKernelGGL(spd_coop_bwd2_kernel<7>, grid, dim3(64), 0, s, a, rounds); return true;
        case 8: hipLaunchKernelGGL(spd_coop_bwd2_kernel<8>, grid, dim3(64), 0, s, a, rounds); return true;
        case 9: hipLaunchKernelGGL(spd_coop_bwd2_kernel<9>, grid, dim3(64), 0, s, a, rounds); return true;
        case 10: hipLaunchKernelGGL(spd_coop_bwd2_kernel<10>, grid, dim3(64), 0, s, a, rounds); return true;
        default: return false;
    }
}
}  // namespace sympa_hip
