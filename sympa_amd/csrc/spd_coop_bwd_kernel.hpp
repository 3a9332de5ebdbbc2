// Kernel of the sixteen-lanes-per-pair SPD backward (routines: spd_coop_bwd.hpp) and its argument block; instantiated
// for M = n in three translation units (spd_bwd.hip: 16, spd_bwd_coop_hi.hip: 12..15, spd_bwd_coop_lo.hip: 3..11) so the
// build stays parallel.
#pragma once

#include "siegel_common.hpp"
#include "spd_coop_bwd.hpp"

namespace sympa_hip {

// Waves per SIMD: two 256-register waves hide each other's dependent chains where the working set fits (M <= 11: no or
// few spills; n = 8 374 -> 341 us, n = 10 539 -> 511 us per 65 536 pairs); at M = 16 the cap spills and is slower
// (1.27 -> 1.64 ms), so the large sizes keep one 512-register wave.
template <int M>
constexpr int spd_coop_bwd_waves() { return M <= 11 ? 2 : 1; }
constexpr int SPD_COOP_BWD_MIN_N = 3;      // below: one lane per pair (spd_bwd_kernel), measured faster

struct SpdBwdArgs {
    const double* x;          // [.., n, n] points or the table
    const double* y;
    const int64_t* src;       // nullptr -> pair i = rows i
    const int64_t* dst;
    int64_t src_stride, dst_stride;
    int64_t num_rows, b;
    const double* scale;
    double inv_scale_coef;
    const double* go;         // [b] or nullptr
    const double* graph_dist; // [b] or nullptr (fused loss)
    double loss_scale;
    double* loss;
    double* gtab;             // scatter form: [num_rows, n, n] table gradient, accumulated with atomics (gx, gy unused); or null
    double* gx;               // [b, n, n] rows of the src / x points
    double* gy;               // [b, n, n] rows of the dst / y points
    double* gscale;
    double* out;
    int32_t* status;
    const int32_t* only_if;   // spd_coop_bwd_kernel: a block runs only when the 64-pair chunk of its first pair is flagged here
                              // (nullptr: every block; a block's pairs must not straddle chunks) -- the last launch behind
                              // the three-kernel backward (spd_coop_bwd3_kernel.hpp) takes the flagged chunks
};

inline dim3 spd_coop_bwd_grid(const int64_t b, const int n) {
    const int rounds = spd_coop::coop_rounds(b, n <= 11 ? 2 : 1);
    return dim3((unsigned)((b + 4 * rounds - 1) / (4 * rounds)));
}

// egrad2rgrad (op 2) / RSGD step (op 1) with sixteen lanes per table row, n = 3..16 (spd_table_coop.hip)
void launch_spd_coop_table(int op, int n, double* x, const double* g, double* out, int64_t b, double lr, double wd,
                           const double* clip, double max_norm, int32_t* status, hipStream_t s);
bool launch_spd_bwd_coop2_lo(const SpdBwdArgs& a, int n, hipStream_t s);             // two rounds per QL, n = 3..10
bool launch_spd_bwd_coop2_hi(const SpdBwdArgs& a, int n, hipStream_t s);             // n = 11..16
void launch_spd_coop_bwd_hi(const SpdBwdArgs& a, int n, dim3 grid, hipStream_t s);   // n = 12..15
void launch_spd_coop_bwd_lo(const SpdBwdArgs& a, int n, dim3 grid, hipStream_t s);   // n = 3..11

// Sixteen lanes per pair (spd_coop_bwd.hpp): one wave per block, `rounds` (spd_coop::coop_rounds) rounds of 4 pairs per
// wave; group g of the wave handles pair 4 (rounds * block + t) + g in round t, lane r of the group owns row r of every
// matrix of that pair.
template <int M>
__global__ __launch_bounds__(64, spd_coop_bwd_waves<M>()) void spd_coop_bwd_kernel(const SpdBwdArgs a, const int rounds) {
    using namespace spd_coop;
    if (a.only_if != nullptr && a.only_if[((int64_t)blockIdx.x * rounds * 4) >> 6] == 0) return;      // block-uniform
    __shared__ __attribute__((aligned(16))) double tbuf_all[4 * TBUF];
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15;
    double* const tbuf = tbuf_all + g * TBUF;
    constexpr int nn = M * M;
    double sc = 1.0;
    bool sc_active = false;
    if (a.scale != nullptr) {
        const double raw = a.scale[0] * a.inv_scale_coef;
        sc_active = raw > 0.1;
        sc = sc_active ? raw : 0.1;
    }
    int st = 0;
    double loss_acc = 0.0, gscale_acc = 0.0;
    for (int t = 0; t < rounds; ++t) {
        const int64_t first = ((int64_t)blockIdx.x * rounds + t) * 4;
        if (first >= a.b) break;                                     // wave-uniform
        const int64_t i = first + g;                                 // my group's pair in this round
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
        const double* py = a.y + r2 * nn;
        double x[M], y[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int lo = r < j ? r : j, hi = r < j ? j : r;
            const int e = (hi < M) ? lo * M + hi : 0;            // upper triangle; a phantom lane reads element 0
            x[j] = px[e];
            y[j] = py[e];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) y[j] -= x[j];                 // D = Y - X
        double rd[M], m[M];
        const bool pd = cholesky_rows(x, rd);                     // x <- rows of L
        solve_right_lt(y, x, rd);                                 // W = D L^-T
        transpose_rows(y, m, tbuf, r);
        solve_right_lt(m, x, rd);                                 // A = L^-1 D L^-T
        double d[M], e[M], vk[M], bk[M];
        tridiagonalize_keep(m, r, d, e, vk, bk);
        // The QL below runs redundantly in the sixteen lanes of the pair and its predicates must agree bit for bit:
        // take (d, e) from lane 0 (they are uniform by construction -- this pins it against compiler reassociation).
#pragma unroll
        for (int j = 0; j < M; ++j) { d[j] = bcast<0>(d[j]); e[j] = bcast<0>(e[j]); }
        double zrow[M];
#pragma unroll
        for (int j = 0; j < M; ++j) zrow[j] = (r == j) ? 1.0 : 0.0;
        const bool conv = tridiag_ql_vectors_row(d, e, zrow);
        double zc[M], vrow[M];
        transpose_rows(zrow, zc, tbuf, r);                        // lane c holds column c of Z
        back_transform_columns(zc, vk, bk);                       // ... of V = Q Z
        transpose_rows(zc, vrow, tbuf, r);                        // lane i holds row i of V
        // distance and the two spectral weights (group-uniform)
        bool ok = pd;
        double acc = 0.0, f[M];
#pragma unroll
        for (int k = 0; k < M; ++k) {
            ok = ok && (d[k] > -1.0);
            f[k] = sympa::d_log1p_signed(d[k]);
            acc = sympa::d_fma(f[k], f[k], acc);
        }
        const double dist = sympa::d_sqrt(acc);
        const double inv = (dist > 0.0) ? sympa::d_rcp(dist) : 0.0;
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
        const double fs = go * sc * inv;
        double gy[M], gx[M];
#pragma unroll
        for (int k = 0; k < M; ++k) {
            gy[k] = fs * f[k] * sympa::d_rcp(1.0 + d[k]);
            gx[k] = -fs * f[k];
        }
        double py_[M], px_[M];
        vdvt_rows(vrow, gy, py_);
        vdvt_rows(vrow, gx, px_);
        congruence_inv_t_rows(py_, x, rd, tbuf, r);
        congruence_inv_t_rows(px_, x, rd, tbuf, r);
        if (a.gtab != nullptr) {            // in-kernel scatter, consecutive lanes on consecutive doubles (spd_coop.hpp)
            scatter_plane<M>(px_, tbuf, a.gtab + r1 * nn, r, live && !bad);
            scatter_plane<M>(py_, tbuf, a.gtab + r2 * nn, r, live && !bad);
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
                if (bad) st |= sympa::ST_BAD_INDEX;
                if (!ok) st |= sympa::ST_NOT_PD;
                if (!conv) st |= sympa::ST_NO_CONVERGENCE;
                if (!sympa::d_finite(dist)) st |= sympa::ST_NONFINITE;
            }
        }
    }
    // lanes r = 0 of the four groups hold the partial sums
    double v = (r == 0) ? loss_acc : 0.0;
    double w = (r == 0) ? gscale_acc : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off); w += __shfl_xor(w, off); }
    if (lane == 0) {
        if (a.loss != nullptr && a.graph_dist != nullptr && v != 0.0) atomicAdd(a.loss, v);
        if (a.gscale != nullptr && a.scale != nullptr && w != 0.0) atomicAdd(a.gscale, w);
    }
    if (a.status != nullptr) {
        const unsigned long long mk = __ballot(st != 0);
        if (mk != 0ull) {
            if (st != 0) atomicOr(&a.status[0], st);
            if (lane == 0) atomicAdd(&a.status[1], (int)__popcll(mk));
        }
    }
}

// The same backward with the QL of TWO rounds run together (tridiag_ql_vectors_2rows): a wave takes 8 pairs per step --
// pair A = first + g and pair B = first + 4 + g for group g.  Everything but the QL is done pair by pair in the
// sixteen-lanes layout (A's factor and reflectors stay in registers while B's are computed); for the QL, lanes 0-7 of a
// group carry the sixteen rows of A's Z, two each, and lanes 8-15 those of B's, so the scalar recurrence -- two thirds
// of the single-round kernel -- is paid once for eight pairs instead of four.  Z goes back to the sixteen-lanes layout
// through LDS, where the single-round kernel transposes it anyway.
template <int M>
__global__ __launch_bounds__(64) void spd_coop_bwd2_kernel(const SpdBwdArgs a, const int rounds) {
    using namespace spd_coop;
    __shared__ __attribute__((aligned(16))) double tbuf_all[4 * TBUF + 2 * 4 * N * N];     // per group: transposes, Z of A, Z of B
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15;
    double* const tbuf = tbuf_all + g * TBUF;
    double* const zt_a = tbuf_all + 4 * TBUF + g * N * N;
    double* const zt_b = tbuf_all + 4 * TBUF + (4 + g) * N * N;
    constexpr int nn = M * M;
    double sc = 1.0;
    bool sc_active = false;
    if (a.scale != nullptr) {
        const double raw = a.scale[0] * a.inv_scale_coef;
        sc_active = raw > 0.1;
        sc = sc_active ? raw : 0.1;
    }
    int st = 0, nflag = 0;
    double loss_acc = 0.0, gscale_acc = 0.0;

    // What a pair keeps across the joint QL: the rows of L, 1 / diag(L), the reflectors and their beta.  SLIM (M >= 14, where
    // two such states and the QL do not fit 512 registers): NORMALISED reflectors (P_k = I - w_k w_k^T, w = sqrt(beta) v: no
    // beta array) and 1 / diag(L) recomputed when the congruences need it -- n = 16 1124 -> 1076 us per 65 536 pairs; for
    // smaller M the extra square roots and reciprocals cost more than the registers (n = 12 500 -> 521 us).
    constexpr bool SLIM = M >= 14;
    struct PairState { double l[M], rd[M], vk[M], bk[M], d[M], e[M]; bool pd, live, bad; int64_t i, r1, r2; };
    // rows in, Cholesky factor + tridiagonal form + reflectors out
    auto front = [&](const int64_t i, PairState& p) {
        p.i = i;
        p.live = i < a.b;
        const int64_t ii = p.live ? i : a.b - 1;
        int64_t r1 = ii, r2 = ii;
        p.bad = false;
        if (a.src != nullptr) {
            r1 = a.src[ii * a.src_stride];
            r2 = a.dst[ii * a.dst_stride];
            if (r1 < 0 || r1 >= a.num_rows || r2 < 0 || r2 >= a.num_rows) { p.bad = true; r1 = 0; r2 = 0; }
        }
        p.r1 = r1; p.r2 = r2;
        const double* px = a.x + r1 * nn;
        const double* py = a.y + r2 * nn;
        double y[M], m[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int lo = r < j ? r : j, hi = r < j ? j : r;
            const int e = (hi < M) ? lo * M + hi : 0;            // upper triangle; a phantom lane reads element 0
            p.l[j] = px[e];
            y[j] = py[e] - p.l[j];                               // D = Y - X
        }
        p.pd = cholesky_rows(p.l, p.rd);                         // l <- rows of L
        solve_right_lt(y, p.l, p.rd);                            // W = D L^-T
        transpose_rows(y, m, tbuf, r);
        solve_right_lt(m, p.l, p.rd);                            // A = L^-1 D L^-T
        tridiagonalize_keep(m, r, p.d, p.e, p.vk, p.bk);
        if constexpr (SLIM) {
#pragma unroll
            for (int j = 0; j < M; ++j) p.vk[j] *= sympa::d_sqrt(p.bk[j]);    // w_k = sqrt(beta_k) v_k; rd, bk are dead from here
        }
#pragma unroll
        for (int j = 0; j < M; ++j) { p.d[j] = bcast<0>(p.d[j]); p.e[j] = bcast<0>(p.e[j]); }     // bitwise uniform (see above)
    };
    // eigenvalues d (group-uniform) and Z (in LDS, row-major 16 x 16) in, gradient rows out
    auto back = [&](PairState& p, const double (&d)[M], const double* __restrict__ zt, const bool conv) {
        double zc[M], vrow[M];
#pragma unroll
        for (int j = 0; j < M; ++j) zc[j] = zt[j * N + r];       // lane c holds column c of Z
        double rd[M];
        if constexpr (SLIM) {
            double one[M];
#pragma unroll
            for (int j = 0; j < M; ++j) one[j] = 1.0;
            back_transform_columns(zc, p.vk, one);               // ... of V = Q Z  (beta folded into the reflectors)
            sfor<0, M>([&](auto J) { rd[J] = sympa::d_rcp(bcast<J>(settle(p.l[J]))); });         // 1 / L[j][j]
        } else {
            back_transform_columns(zc, p.vk, p.bk);              // ... of V = Q Z
#pragma unroll
            for (int j = 0; j < M; ++j) rd[j] = p.rd[j];
        }
        transpose_rows(zc, vrow, tbuf, r);                       // lane i holds row i of V
        bool ok = p.pd;
        double acc = 0.0, f[M];
#pragma unroll
        for (int k = 0; k < M; ++k) {
            ok = ok && (d[k] > -1.0);
            f[k] = sympa::d_log1p_signed(d[k]);
            acc = sympa::d_fma(f[k], f[k], acc);
        }
        const double dist = sympa::d_sqrt(acc);
        const double inv = (dist > 0.0) ? sympa::d_rcp(dist) : 0.0;
        double go = 0.0, loss_i = 0.0;
        if (a.graph_dist != nullptr) {
            const double gd = p.live ? a.graph_dist[p.i] : 1.0;
            const double ratio = dist * sc / gd;
            const double ee = ratio * ratio - 1.0;
            loss_i = (p.live && !p.bad) ? fabs(ee) * a.loss_scale : 0.0;
            go = (ee > 0.0 ? 1.0 : (ee < 0.0 ? -1.0 : 0.0)) * 2.0 * ratio / gd * a.loss_scale;
        } else if (a.go != nullptr) {
            go = p.live ? a.go[p.i] : 0.0;
        }
        if (!p.live || p.bad) go = 0.0;
        const double fs = go * sc * inv;
        double gy[M], gx[M];
#pragma unroll
        for (int k = 0; k < M; ++k) {
            gy[k] = fs * f[k] * sympa::d_rcp(1.0 + d[k]);
            gx[k] = -fs * f[k];
        }
        double py_[M], px_[M];
        vdvt_rows(vrow, gy, py_);
        vdvt_rows(vrow, gx, px_);
        congruence_inv_t_rows(py_, p.l, rd, tbuf, r);
        congruence_inv_t_rows(px_, p.l, rd, tbuf, r);
        if (a.gtab != nullptr) {
            scatter_plane<M>(px_, tbuf, a.gtab + p.r1 * nn, r, p.live && !p.bad);
            scatter_plane<M>(py_, tbuf, a.gtab + p.r2 * nn, r, p.live && !p.bad);
        } else if (p.live && r < M) {
            double* ox = a.gx + p.i * nn + r * M;
            double* oy = a.gy + p.i * nn + r * M;
#pragma unroll
            for (int j = 0; j < M; ++j) { ox[j] = px_[j]; oy[j] = py_[j]; }
        }
        if (r == 0) {
            if (p.live && a.out != nullptr) a.out[p.i] = p.bad ? __builtin_nan("") : dist * sc;
            loss_acc += loss_i;
            gscale_acc += (p.live && sc_active) ? go * dist * a.inv_scale_coef : 0.0;
            if (p.live) {
                int s = 0;
                if (p.bad) s |= sympa::ST_BAD_INDEX;
                if (!ok) s |= sympa::ST_NOT_PD;
                if (!conv) s |= sympa::ST_NO_CONVERGENCE;
                if (!sympa::d_finite(dist)) s |= sympa::ST_NONFINITE;
                st |= s;
                nflag += (s != 0) ? 1 : 0;
            }
        }
    };

    for (int t = 0; t < rounds; ++t) {
        const int64_t first = ((int64_t)blockIdx.x * rounds + t) * 8;
        if (first >= a.b) break;                                     // wave-uniform
        PairState pa, pb;
        front(first + g, pa);
        front(first + 4 + g, pb);
        // joint QL: lanes 0-7 of the group take A's tridiagonal, lanes 8-15 B's; rows 2 (r & 7) and 2 (r & 7) + 1 of Z
        const bool half_b = r >= 8;
        double d[M], e[M], z0[M], z1[M];
        const int row0 = 2 * (r & 7);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            d[j] = half_b ? pb.d[j] : pa.d[j];
            e[j] = half_b ? pb.e[j] : pa.e[j];
            z0[j] = (j == row0) ? 1.0 : 0.0;
            z1[j] = (j == row0 + 1) ? 1.0 : 0.0;
        }
        const bool conv = tridiag_ql_vectors_2rows(d, e, z0, z1);
        wave_lds_fence();
        {
            double* zt = half_b ? zt_b : zt_a;
#pragma unroll
            for (int j = 0; j < M; ++j) { zt[row0 * N + j] = z0[j]; zt[(row0 + 1) * N + j] = z1[j]; }
        }
        wave_lds_fence();
        const bool conv_a = __shfl((int)conv, lane & 48) != 0, conv_b = __shfl((int)conv, (lane & 48) + 8) != 0;
        double da[M], db[M];
#pragma unroll
        for (int j = 0; j < M; ++j) { da[j] = bcast<0>(d[j]); db[j] = bcast<8>(d[j]); }
        back(pa, da, zt_a, conv_a);
        back(pb, db, zt_b, conv_b);
    }
    // lanes r = 0 of the four groups hold the partial sums
    double v = (r == 0) ? loss_acc : 0.0;
    double w = (r == 0) ? gscale_acc : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off); w += __shfl_xor(w, off); }
    if (lane == 0) {
        if (a.loss != nullptr && a.graph_dist != nullptr && v != 0.0) atomicAdd(a.loss, v);
        if (a.gscale != nullptr && a.scale != nullptr && w != 0.0) atomicAdd(a.gscale, w);
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

}  // namespace sympa_hip
