// SPD backward in THREE kernels (round 4): the eigen-decomposition of A = L^-1 (Y - X) L^-T no longer runs the QL iteration
// with accumulated rotations in the sixteen-lanes layout -- where the scalar recurrence of a pair is executed redundantly by
// its sixteen lanes, four (eight) pairs per instruction stream, 80 % of the old kernel -- but
//
//   kernel A  (sixteen lanes per pair, 4 pairs per round): rows in, Cholesky, A, Householder tridiagonal form with the
//             reflectors kept -- spd_coop_bwd.hpp, unchanged -- reflectors, betas and (d, e) -> workspace slot of the pair
//   kernel B  (ONE PAIR PER LANE, 64 pairs per instruction stream): eigenvalues by the forward's lockstep PWK QL, sorted;
//             eigenvectors of T by inverse iteration (tridiag_invit.hpp: LU with partial pivoting of T - lambda I, two solves,
//             Gram-Schmidt inside clusters); distance and the spectral weights once per pair; all of it -> workspace
//   kernel C  (sixteen lanes per pair again): X rows in, Cholesky once more (cheaper than carrying L through the workspace),
//             Z and the reflectors back from the workspace, V = Q Z, P = V diag(g) V^T, L^-T P L^-1, loss, scatter -- the tail of
//             the old kernel, unchanged.
//
// Three launches instead of one kernel with three phases because the phases want different register budgets: B holds a
// tridiagonal LU, three previous vectors and the iterate of its pair (one 512-register wave per SIMD), A and C fit 256 registers
// and run two waves per SIMD, which is worth 1.5x on these latency-bound dependent chains (measured: one fused kernel 0.61 ms
// per 65 536 pairs, see profiles/r04_spd_backward.txt).  The workspace is caller-owned (sympa_spd_backward_workspace_bytes).
// A pair whose spectrum has a block of more than INVIT_KEEP + 1 close eigenvalues (y = c x: all equal) cannot be served by
// the register-resident Gram-Schmidt: kernel B flags its 64-pair CHUNK, kernel C skips flagged chunks, and a last launch of the
// QL-with-vectors kernel restricted to flagged chunks (an early exit everywhere else) finishes them.
#pragma once

#include "spd_coop_bwd_kernel.hpp"
#include "tridiag_invit.hpp"

namespace sympa_hip {

// per pair slot (index = pair index): reflector components PACKED (lane r holds component r of reflectors k < r, nothing else is
// non-zero: r (r - 1) / 2 + k), betas [M], tridiagonal form d [M], e [M], the
// spectral weights F_k = log(1 + a_k) / dist and H_k = F_k / (1 + a_k) [M each] (computed ONCE per pair in kernel B instead of
// redundantly by sixteen lanes), {dist, flags: bit 0 "every 1 + a_k > 0", bit 1 "QL converged"}, eigenvectors [M][M]
template <int M>
struct SpdBwd3Ws {
    static constexpr int64_t VK = 0;
    static constexpr int64_t BK = VK + M * (M - 1) / 2;
    static constexpr int64_t D = BK + M;
    static constexpr int64_t E = D + M;
    static constexpr int64_t F = E + M;
    static constexpr int64_t H = F + M;
    static constexpr int64_t DIST = H + M;
    static constexpr int64_t Z = DIST + 2;
    static constexpr int64_t SLOT = Z + M * M;          // doubles per pair slot
};
inline int64_t spd_bwd3_slot_doubles(int n) { return (int64_t)n * (n - 1) / 2 + 5 * n + 2 + (int64_t)n * n; }
// workspace: [chunks of 64 pairs] int32 flags (8 bytes each, 16-byte aligned total), then one slot per pair (whole chunks)
inline int64_t spd_bwd3_workspace_bytes(int64_t b, int n) {
    const int64_t chunks = (b + 63) / 64;
    return 8 * (chunks + (chunks & 1)) + chunks * 64 * spd_bwd3_slot_doubles(n) * 8;
}
// rounds per wave of kernels A and C: a power of two <= 16 (a block's pairs never straddle a 64-pair chunk), small enough
// that the batch spreads over all SIMDs at two waves each
inline int spd_bwd3_rounds(int64_t b) {
    const int want = spd_coop::coop_rounds(b, 2);
    int r = 1;
    while (2 * r <= want && 2 * r <= 16) r *= 2;
    return r;
}

template <int M>
__global__ __launch_bounds__(64, 2) void spd_bwd3_front_kernel(const SpdBwdArgs a, const int rounds, double* __restrict__ ws) {
    using namespace spd_coop;
    using W = SpdBwd3Ws<M>;
    __shared__ __attribute__((aligned(16))) double tbuf_all[4 * TBUF];
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15;
    double* const tbuf = tbuf_all + g * TBUF;
    constexpr int nn = M * M;
    for (int t = 0; t < rounds; ++t) {
        const int64_t first = ((int64_t)blockIdx.x * rounds + t) * 4;
        if (first >= a.b) break;                                     // wave-uniform
        const int64_t i = first + g;                                 // slots exist for whole chunks of 64: i < chunks * 64
        const int64_t ii = i < a.b ? i : a.b - 1;
        int64_t r1 = ii, r2 = ii;
        if (a.src != nullptr) {
            r1 = a.src[ii * a.src_stride];
            r2 = a.dst[ii * a.dst_stride];
            if (r1 < 0 || r1 >= a.num_rows || r2 < 0 || r2 >= a.num_rows) { r1 = 0; r2 = 0; }
        }
        const double* px = a.x + r1 * nn;
        const double* py = a.y + r2 * nn;
        double l[M], rd[M], y[M], m[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int lo = r < j ? r : j, hi = r < j ? j : r;
            const int e = (hi < M) ? lo * M + hi : 0;            // upper triangle; a phantom lane reads element 0
            l[j] = px[e];
            y[j] = py[e] - l[j];                                 // D = Y - X
        }
        cholesky_rows(l, rd);                                    // l <- rows of L
        solve_right_lt(y, l, rd);                                // W = D L^-T
        transpose_rows(y, m, tbuf, r);
        solve_right_lt(m, l, rd);                                // A = L^-1 D L^-T
        double d[M], e[M], vk[M], bk[M];
        tridiagonalize_keep(m, r, d, e, vk, bk);
        double* const slot = ws + i * W::SLOT;
        {
            double* const mine = slot + W::VK + r * (r - 1) / 2;
#pragma unroll
            for (int j = 0; j < M - 1; ++j)
                if (j < r && r < M) mine[j] = vk[j];
        }
        // group-uniform values (every lane holds all of them): lane r stores entry r -- three 128-byte stores per pair instead of
        // 48 eight-byte ones from lane 0
        if (r < M) {
            double bkr = bk[0], dr = d[0], er = e[0];
#pragma unroll
            for (int j = 1; j < M; ++j) { bkr = (r == j) ? bk[j] : bkr; dr = (r == j) ? d[j] : dr; er = (r == j) ? e[j] : er; }
            slot[W::BK + r] = bkr;
            slot[W::D + r] = dr;
            slot[W::E + r] = er;
        }
    }
}

// one pair per lane: pair = blockIdx.x * 64 + lane (chunk = block)
template <int M>
__global__ __launch_bounds__(64) void spd_bwd3_eig_kernel(const int64_t b, double* __restrict__ ws, int32_t* __restrict__ chunk_flags) {
    using W = SpdBwd3Ws<M>;
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool live = i < b;
    double* const slot = ws + (live ? i : b - 1) * W::SLOT;      // tail lanes redo the last pair (the lockstep QL wants valid data)
    double dk[M], ek[M], w[M], e2[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        dk[j] = slot[W::D + j];
        ek[j] = slot[W::E + j];
        w[j] = dk[j];
        e2[j] = (j < M - 1) ? ek[j] * ek[j] : 0.0;
    }
    const bool conv = sympa::tridiag_ql_lockstep<M>(w, e2);
    sympa::sort_ascending<M>(w);
    double* const out = ws + i * W::SLOT;                        // (a dead lane's slot exists: whole chunks are allocated)
    // (the eigenvectors leave lane by lane, M eight-byte stores of 64 different lines each: staging a vector of all 64 pairs in the
    // LDS and writing whole 128-byte vectors was measured -- 1.49 -> 1.60 ms per 1 M pairs: the lone wave waits for its own LDS round
    // trip per vector, the stores it saves were overlapped anyway)
    bool small_blocks = true;
    small_blocks = sympa::tridiag_eigvecs_invit<M>(dk, ek, w, [&](auto IC, const double (&x)[M]) {
        constexpr int k = decltype(IC)::value;
#pragma unroll
        for (int j = 0; j < M; ++j) out[W::Z + k * M + j] = x[j];
    });
    // distance and spectral weights of MY pair (one lane, once): kernel C only scales them with the loss gradient
    bool ok = true;
    double acc = 0.0, f[M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        ok = ok && (w[k] > -1.0);
        f[k] = sympa::d_log1p_signed(w[k]);
        acc = sympa::d_fma(f[k], f[k], acc);
    }
    const double dist = sympa::d_sqrt(acc);
    const double inv = (dist > 0.0) ? sympa::d_rcp(dist) : 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        const double fk = f[k] * inv;
        out[W::F + k] = fk;
        out[W::H + k] = fk * sympa::d_rcp(1.0 + w[k]);
    }
    out[W::DIST] = dist;
    out[W::DIST + 1] = (ok ? 1.0 : 0.0) + (conv ? 2.0 : 0.0);
    // (dist = 0, y = x: every weight F_k, H_k is zero, so the vectors do not matter -- no reason to hand the chunk back)
    const bool fallback = __ballot(live && !small_blocks && dist > 0.0) != 0ull;
    if (lane == 0) chunk_flags[blockIdx.x] = fallback ? 1 : 0;
}

// one wave per SIMD: at 256 registers it spills 77 of them (8.26 -> 8.13 ms per 1 M pairs with one wave; the kernel is HBM-bound)
template <int M>
__global__ __launch_bounds__(64, 1) void spd_bwd3_back_kernel(const SpdBwdArgs a, const int rounds, const double* __restrict__ ws,
                                                              const int32_t* __restrict__ chunk_flags) {
    using namespace spd_coop;
    using W = SpdBwd3Ws<M>;
    __shared__ __attribute__((aligned(16))) double tbuf_all[4 * TBUF];
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15;
    double* const tbuf = tbuf_all + g * TBUF;
    constexpr int nn = M * M;
    // my block's pairs lie inside ONE chunk (rounds is a power of two <= 16): flagged chunks belong to the QL kernel
    if (chunk_flags[((int64_t)blockIdx.x * rounds * 4) >> 6] != 0) return;
    double sc = 1.0;
    bool sc_active = false;
    if (a.scale != nullptr) {
        const double raw = a.scale[0] * a.inv_scale_coef;
        sc_active = raw > 0.1;
        sc = sc_active ? raw : 0.1;
    }
    int st = 0, nflag = 0;
    double loss_acc = 0.0, gscale_acc = 0.0;
    // the pending source-row gradient of the wave (see the scatter below): nn doubles, lane l owns l, l + 64, ...
    constexpr int PEND_PER_LANE = (nn + 63) / 64;
    __shared__ double pend[PEND_PER_LANE * 64];
    int pend_row = -1;
    auto pend_flush = [&](const int row) {
        double* base = a.gtab + (int64_t)row * nn;
#pragma unroll
        for (int k = 0; k < PEND_PER_LANE; ++k) {
            const int idx = lane + 64 * k;
            const double v = (idx < nn) ? pend[idx] : 0.0;
            if (idx < nn && v != 0.0) atomicAdd(base + idx, v);
        }
    };
    for (int t = 0; t < rounds; ++t) {
        const int64_t first = ((int64_t)blockIdx.x * rounds + t) * 4;
        if (first >= a.b) break;
        const int64_t i = first + g;
        const bool live = i < a.b;
        const int64_t ii = live ? i : a.b - 1;
        int64_t r1 = ii, r2 = ii;
        bool bad = false;
        if (a.src != nullptr) {
            r1 = a.src[ii * a.src_stride];
            r2 = a.dst[ii * a.dst_stride];
            if (r1 < 0 || r1 >= a.num_rows || r2 < 0 || r2 >= a.num_rows) { bad = true; r1 = 0; r2 = 0; }
        }
        const double* px = a.x + r1 * nn;
        const double* const slot = ws + ii * W::SLOT;
        // everything this round reads is requested up front; the Cholesky below runs while the workspace loads are in flight
        double l[M], rd[M], vk[M], bk[M], fw[M], hw[M], zc[M], vrow[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int lo = r < j ? r : j, hi = r < j ? j : r;
            const int e = (hi < M) ? lo * M + hi : 0;
            l[j] = px[e];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            vk[j] = (j < r && r < M) ? slot[W::VK + r * (r - 1) / 2 + j] : 0.0;
            zc[j] = (r < M) ? slot[W::Z + r * M + j] : 0.0;      // lane c holds eigenvector c of T = column c of Z
            bk[j] = slot[W::BK + j];
            fw[j] = slot[W::F + j];
            hw[j] = slot[W::H + j];
        }
        const double dist = slot[W::DIST];
        const int eig_flags = (int)slot[W::DIST + 1];
        const bool pd = cholesky_rows(l, rd);
        const bool ok = pd && (eig_flags & 1);
        const bool conv = (eig_flags & 2) != 0;
        back_transform_columns(zc, vk, bk);                      // ... column c of V = Q Z
        transpose_rows(zc, vrow, tbuf, r);                       // lane i holds row i of V
        double go = 0.0, loss_i = 0.0;
        if (a.graph_dist != nullptr) {
            const double gd = live ? a.graph_dist[i] : 1.0;
            const double ratio = dist * sc / gd;
            const double ee = ratio * ratio - 1.0;
            loss_i = (live && !bad) ? fabs(ee) * a.loss_scale : 0.0;
            go = (ee > 0.0 ? 1.0 : (ee < 0.0 ? -1.0 : 0.0)) * 2.0 * ratio / gd * a.loss_scale;
        } else if (a.go != nullptr) {
            go = live ? a.go[i] : 0.0;
        }
        if (!live || bad) go = 0.0;
        const double fs = go * sc;
        double gy[M], gx[M];
#pragma unroll
        for (int k = 0; k < M; ++k) {
            gy[k] = fs * hw[k];
            gx[k] = -fs * fw[k];
        }
        double py_[M], px_[M];
        vdvt_rows(vrow, gy, py_);
        vdvt_rows(vrow, gx, px_);
        congruence_inv_t_rows(py_, l, rd, tbuf, r);
        congruence_inv_t_rows(px_, l, rd, tbuf, r);
        if (a.gtab != nullptr) {
            // SOURCE side through the wave's pending tile: consecutive pairs with the same source row (a batch sorted by its first
            // column: sympa_amd/data.py::sort_batches_by_source, ~10 pairs per source row at configs[4]) are summed in the LDS and
            // leave as ONE run of full-width atomic instructions when the row changes -- the memory side retires fp64 atomics by
            // the 128-byte line (tools/microbench/atomic_rate.hip), and the scatter was 4.3 of the 8.6 GB this kernel moves per
            // 1 M pairs.  The four pairs of a round are taken in order (wave-uniform loop, rows are scalars); dead pairs repeat the
            // last live row with zeros.
            const bool on = live && !bad;
            wave_lds_fence();
            if (r < M) sfor<0, M>([&](auto J) { tbuf[r * M + J] = on ? px_[J] : 0.0; });
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row_q = __builtin_amdgcn_readlane((int)r1, q * 16);
                const double* tq = tbuf_all + q * TBUF;
                pend_row = __builtin_amdgcn_readfirstlane(pend_row);         // a scalar, and the compiler shall know it
                if (row_q != pend_row) {
                    if (pend_row >= 0) pend_flush(pend_row);
                    pend_row = row_q;
#pragma unroll
                    for (int k = 0; k < PEND_PER_LANE; ++k)
                        if (lane + 64 * k < nn) pend[lane + 64 * k] = tq[lane + 64 * k];
                } else {
#pragma unroll
                    for (int k = 0; k < PEND_PER_LANE; ++k)
                        if (lane + 64 * k < nn) pend[lane + 64 * k] += tq[lane + 64 * k];
                }
            }
            scatter_plane<M>(py_, tbuf, a.gtab + r2 * nn, r, on);
        } else if (live && r < M) {
            double* ox = a.gx + i * nn + r * M;
            double* oy = a.gy + i * nn + r * M;
#pragma unroll
            for (int j = 0; j < M; ++j) { ox[j] = px_[j]; oy[j] = py_[j]; }
        }
        if (r == 0) {
            if (live && a.out != nullptr) a.out[i] = bad ? __builtin_nan("") : dist * sc;
            loss_acc += loss_i;
            gscale_acc += (live && sc_active) ? go * dist * a.inv_scale_coef : 0.0;
            if (live) {
                int s = 0;
                if (bad) s |= sympa::ST_BAD_INDEX;
                if (!ok) s |= sympa::ST_NOT_PD;
                if (!conv) s |= sympa::ST_NO_CONVERGENCE;
                if (!sympa::d_finite(dist)) s |= sympa::ST_NONFINITE;
                st |= s;
                nflag += (s != 0) ? 1 : 0;
            }
        }
    }
    pend_row = __builtin_amdgcn_readfirstlane(pend_row);
    if (a.gtab != nullptr && pend_row >= 0) {
        wave_lds_fence();
        pend_flush(pend_row);
    }
    double v = (r == 0) ? loss_acc : 0.0;
    double w2 = (r == 0) ? gscale_acc : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off); w2 += __shfl_xor(w2, off); }
    if (lane == 0) {
        if (a.loss != nullptr && a.graph_dist != nullptr && v != 0.0) atomicAdd(a.loss, v);
        if (a.gscale != nullptr && a.scale != nullptr && w2 != 0.0) atomicAdd(a.gscale, w2);
    }
    if (a.status != nullptr) {
        if (__ballot(st != 0) != 0ull) {
            if (st != 0) atomicOr(&a.status[0], st);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nflag += __shfl_xor(nflag, off);
            if (lane == 0) atomicAdd(&a.status[1], nflag);
        }
    }
}

// launches the three kernels and the restricted QL-with-vectors kernel behind them (spd_bwd3.hip); false: n not instantiated
bool launch_spd_bwd3(const SpdBwdArgs& a, int n, void* workspace, int64_t workspace_bytes, hipStream_t s, int* rc);

}  // namespace sympa_hip
