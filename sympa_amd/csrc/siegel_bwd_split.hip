// Split Siegel backward (dims 5..8, one pair per lane, two kernels through a caller-owned workspace): workspace size and dispatch.
// Kernels: siegel_bwd_split_kernel.hpp, one per translation unit (siegel_bwd_split_spectral_*.hip, siegel_bwd_split_gradient_*.hip).
#include "siegel_bwd_split_kernel.hpp"

namespace sympa_hip {
int launch_bwd_split_spectral_upper_5(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_5_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_5_dense(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_spectral_upper_6(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_6_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_6_dense(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_spectral_upper_7(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_7_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_7_dense(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_spectral_upper_8(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_8_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_upper_8_dense(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_spectral_bounded_5(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_5_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_5_dense(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_spectral_bounded_6(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_6_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_6_dense(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_spectral_bounded_7(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_7_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_7_dense(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_spectral_bounded_8(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_8_scatter(const SplitArgs& sa, hipStream_t s);
int launch_bwd_split_gradient_bounded_8_dense(const SplitArgs& sa, hipStream_t s);

namespace {
int pack_len(int n, int model) {
    const int offd = n * (n - 1) / 2;
    return 2 * n + 3 * offd + (model == SYMPA_MODEL_UPPER ? 0 : offd);            // sympa::AdjPack<n, model>::LEN
}
// behind the packs: one word per wave of 64 pairs (SplitArgs::graded: 1 = stage 1 hands the wave to the one-stage kernel), rounded up
// to 16 bytes
int64_t flag_bytes(int64_t b) { return (((b + 63) / 64) * (int64_t)sizeof(int) + 15) / 16 * 16; }
int64_t padded(int64_t b) { return (b + 63) / 64 * 64; }
}  // namespace

bool bwd_split_available(int n, int model) {
    return n >= 5 && n <= 8 && (model == SYMPA_MODEL_UPPER || model == SYMPA_MODEL_BOUNDED);
}

int64_t bwd_split_workspace_bytes(int64_t b, int n, int model) {
    if (!bwd_split_available(n, model) || b <= 0) return 0;
    return (int64_t)pack_len(n, model) * padded(b) * (int64_t)sizeof(double) + flag_bytes(b);
}

int launch_bwd_split(const BwdArgs& a, int n, int model, bool scatter, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    if (!bwd_split_available(n, model)) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "split backward: dims 5..8");
    if (workspace == nullptr || workspace_bytes < bwd_split_workspace_bytes(a.f.b, n, model))
        return fail(SYMPA_ERR_BAD_ARG, "split backward: workspace smaller than sympa_siegel_backward_workspace_bytes(b, n, model)");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return fail(SYMPA_ERR_BAD_ARG, "split backward: workspace must be 16-byte aligned");
    // the kernels address a pair inside a workspace entry with a 32-bit byte offset
    if (padded(a.f.b) >= ((int64_t)1 << 29)) return fail(SYMPA_ERR_BAD_ARG, "split backward: at most 2^29 - 64 pairs per call");
    static_assert(sympa::AdjPack<8, sympa::MODEL_UPPER>::LEN == 2 * 8 + 3 * 28, "pack_len out of step with AdjPack");
    static_assert(sympa::AdjPack<5, sympa::MODEL_BOUNDED>::LEN == 2 * 5 + 4 * 10, "pack_len out of step with AdjPack");
    SplitArgs sa;
    sa.a = a;
    sa.ws = static_cast<double*>(workspace);
    sa.ws_stride = padded(a.f.b);
    sa.graded = reinterpret_cast<int*>(static_cast<char*>(workspace) + (int64_t)pack_len(n, model) * padded(a.f.b) * (int64_t)sizeof(double));
    // the spectral kernel staggers its first round of gathers like the dims 7, 8 forward: two rounds or more of a table beyond the L2s
    sa.a.f.flags &= ~SYMPA_INTERNAL_FLAG_STAGGER;
    if (a.f.idx1 != nullptr && a.f.b >= 2048 * 64 && a.f.num_rows * (int64_t)(16 * n * n) >= ((int64_t)12 << 20))
        sa.a.f.flags |= SYMPA_INTERNAL_FLAG_STAGGER;
    const bool upper = model == SYMPA_MODEL_UPPER;
    int rc;
    switch (n) {
        case 5: rc = upper ? launch_bwd_split_spectral_upper_5(sa, s) : launch_bwd_split_spectral_bounded_5(sa, s); break;
        case 6: rc = upper ? launch_bwd_split_spectral_upper_6(sa, s) : launch_bwd_split_spectral_bounded_6(sa, s); break;
        case 7: rc = upper ? launch_bwd_split_spectral_upper_7(sa, s) : launch_bwd_split_spectral_bounded_7(sa, s); break;
        case 8: rc = upper ? launch_bwd_split_spectral_upper_8(sa, s) : launch_bwd_split_spectral_bounded_8(sa, s); break;
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "split backward: dims 5..8");
    }
    if (rc != 0) return rc;
    const auto gradient = [&]() -> int {
    switch (n) {
        case 5:
            if (upper) return scatter ? launch_bwd_split_gradient_upper_5_scatter(sa, s) : launch_bwd_split_gradient_upper_5_dense(sa, s);
            return scatter ? launch_bwd_split_gradient_bounded_5_scatter(sa, s) : launch_bwd_split_gradient_bounded_5_dense(sa, s);
        case 6:
            if (upper) return scatter ? launch_bwd_split_gradient_upper_6_scatter(sa, s) : launch_bwd_split_gradient_upper_6_dense(sa, s);
            return scatter ? launch_bwd_split_gradient_bounded_6_scatter(sa, s) : launch_bwd_split_gradient_bounded_6_dense(sa, s);
        case 7:
            if (upper) return scatter ? launch_bwd_split_gradient_upper_7_scatter(sa, s) : launch_bwd_split_gradient_upper_7_dense(sa, s);
            return scatter ? launch_bwd_split_gradient_bounded_7_scatter(sa, s) : launch_bwd_split_gradient_bounded_7_dense(sa, s);
        case 8:
            if (upper) return scatter ? launch_bwd_split_gradient_upper_8_scatter(sa, s) : launch_bwd_split_gradient_upper_8_dense(sa, s);
            return scatter ? launch_bwd_split_gradient_bounded_8_scatter(sa, s) : launch_bwd_split_gradient_bounded_8_dense(sa, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "split backward: dims 5..8");
    }
    };
    rc = gradient();
    if (rc != 0) return rc;
    // the waves stage 1 listed (graded spectra: siegel_math_bwd_split.hpp) through the one-stage kernel, which refines every
    // eigenvalue to a Rayleigh quotient: a fixed grid of 64 waves that scans the words (usually all zero).  Their packs were zero, so the
    // gradient kernel added / wrote zeros for them; loss, forward values, scale / weight gradients, the deterministic per-wave
    // sums and the status of those waves come from this launch.
    BwdArgs fin = a;
    fin.chunk_flags = sa.graded;
    fin.f.flags &= ~(SYMPA_FLAG_SPLIT | SYMPA_FLAG_COOP | SYMPA_INTERNAL_FLAG_STAGGER);
    return launch_bwd_one_lane(fin, n, model, scatter, s);
}

}  // namespace sympa_hip
