// spd backward, sixteen lanes per pair with the QL of two rounds run together (spd_coop_bwd2_kernel): M = 11..16
#include "spd_coop_bwd_kernel.hpp"

namespace sympa_hip {
bool launch_spd_bwd_coop2_hi(const SpdBwdArgs& a, int n, hipStream_t s) {
    const int rounds = spd_coop::coop_rounds(a.b, 1, 8);
    const dim3 grid((unsigned)((a.b + 8 * rounds - 1) / (8 * rounds)));
    switch (n) {
        case 11: hipLaunchKernelGGL(spd_coop_bwd2_kernel<11>, grid, dim3(64), 0, s, a, rounds); return true;
        case 12: hipLaunchKernelGGL(spd_coop_bwd2_kernel<12>, grid, dim3(64), 0, s, a, rounds); return true;
        case 13: hipLaunchKernelGGL(spd_coop_bwd2_kernel<13>, grid, dim3(64), 0, s, a, rounds); return true;
        case 14: hipLaunchKernelGGL(spd_coop_bwd2_kernel<14>, grid, dim3(64), 0, s, a, rounds); return true;
        case 15: hipLaunchKernelGGL(spd_coop_bwd2_kernel<15>, grid, dim3(64), 0, s, a, rounds); return true;
        case 16: hipLaunchKernelGGL(spd_coop_bwd2_kernel<16>, grid, dim3(64), 0, s, a, rounds); return true;
        default: return false;
    }
}
}  // namespace sympa_hip
