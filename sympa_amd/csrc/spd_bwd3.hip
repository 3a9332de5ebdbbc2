// Three-kernel SPD backward (spd_coop_bwd3_kernel.hpp): instantiations n = 13..16 and the launcher (n = 9..12: spd_bwd3_lo.hip,
// the same source with SYMPA_BWD3_LO, so that the build compiles the two halves in parallel).
#include "spd_coop_bwd3_kernel.hpp"

namespace sympa_hip {

namespace {
template <int M>
int launch3(const SpdBwdArgs& a, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    if (workspace == nullptr || workspace_bytes < spd_bwd3_workspace_bytes(a.b, M))
        return fail(SYMPA_ERR_BAD_ARG, "spd backward: workspace too small (sympa_spd_backward_workspace_bytes)");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return fail(SYMPA_ERR_BAD_ARG, "spd backward: workspace must be 16-byte aligned");
    const int64_t chunks = (a.b + 63) / 64;
    int32_t* flags = reinterpret_cast<int32_t*>(workspace);
    double* ws = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 8 * (chunks + (chunks & 1)));
    const int rounds = spd_bwd3_rounds(a.b);
    const unsigned grid = (unsigned)((a.b + 4 * rounds - 1) / (4 * rounds));
    hipLaunchKernelGGL(spd_bwd3_front_kernel<M>, dim3(grid), dim3(64), 0, s, a, rounds, ws);
    hipLaunchKernelGGL(spd_bwd3_eig_kernel<M>, dim3((unsigned)chunks), dim3(64), 0, s, a.b, ws, flags);
    hipLaunchKernelGGL(spd_bwd3_back_kernel<M>, dim3(grid), dim3(64), 0, s, a, rounds, ws, flags);
    // the pairs of flagged chunks (a block of more than INVIT_KEEP + 1 close eigenvalues): the QL-with-vectors kernel, ONE round per
    // wave (a flagged chunk becomes sixteen waves of 75 us instead of one wave of 1.2 ms at the tail of the step); every other
    // block leaves at once
    SpdBwdArgs m = a;
    m.only_if = flags;
    hipLaunchKernelGGL(spd_coop_bwd_kernel<M>, dim3((unsigned)((a.b + 3) / 4)), dim3(64), 0, s, m, 1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}
}  // namespace

#ifndef SYMPA_BWD3_LO
bool launch_spd_bwd3_lo(const SpdBwdArgs& a, int n, void* workspace, int64_t workspace_bytes, hipStream_t s, int* rc);
bool launch_spd_bwd3(const SpdBwdArgs& a, int n, void* workspace, int64_t workspace_bytes, hipStream_t s, int* rc) {
    switch (n) {
        case 13: *rc = launch3<13>(a, workspace, workspace_bytes, s); return true;
        case 14: *rc = launch3<14>(a, workspace, workspace_bytes, s); return true;
        case 15: *rc = launch3<15>(a, workspace, workspace_bytes, s); return true;
        case 16: *rc = launch3<16>(a, workspace, workspace_bytes, s); return true;
        default: return launch_spd_bwd3_lo(a, n, workspace, workspace_bytes, s, rc);
    }
}
#else
bool launch_spd_bwd3_lo(const SpdBwdArgs& a, int n, void* workspace, int64_t workspace_bytes, hipStream_t s, int* rc) {
    switch (n) {
        case 9: *rc = launch3<9>(a, workspace, workspace_bytes, s); return true;
        case 10: *rc = launch3<10>(a, workspace, workspace_bytes, s); return true;
        case 11: *rc = launch3<11>(a, workspace, workspace_bytes, s); return true;
        case 12: *rc = launch3<12>(a, workspace, workspace_bytes, s); return true;
        default: return false;
    }
}
#endif

}  // namespace sympa_hip
