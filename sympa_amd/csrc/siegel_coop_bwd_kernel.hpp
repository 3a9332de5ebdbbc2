// Kernel of the sixteen-lanes-per-pair Siegel backward (routines: siegel_coop_bwd.hpp), both models, instantiated per
// model, matrix size M = n and output form by siegel_bwd_coop_*.hip.  Same argument block, outputs and fused AverageDistortionLoss as the
// one-pair-per-lane kernel (siegel_bwd_kernel.hpp); SCATTER: atomic adds of the gradient rows into the table gradient,
// otherwise per-pair rows [b, 2, n, n].
#pragma once

#include "siegel_bwd_kernel.hpp"
#include "siegel_coop_bwd.hpp"

namespace sympa_hip {

// One wave per block, `rounds` rounds of 4 pairs per wave: group g of the wave handles pair 4 (rounds * block + t) + g in
// round t and lane r of the group owns row r of every matrix of that pair.  Lanes r >= M are phantoms (spd_coop.hpp).
// The host picks rounds = clamp(b / 4096, 1, 16) (spd_coop::coop_rounds): a 512-register wave owns its SIMD, so a batch
// below 65 536 pairs is spread over all 1024 SIMDs instead of filling a quarter of them with 16-round waves.
// Waves per SIMD (measured per 65 536 pairs, one 512-register wave against two 256-register waves that spill 300-1300
// registers): upper n = 10 836 / 953 us, n = 16 2489 / 3015 us -> one wave;  bounded n = 10 1469 / 1285 us, n = 16
// 4558 / 4036 us -> two waves (its tail of complex solves and products has the longer dependent chains to hide).
template <int MODEL>
constexpr int coop_bwd_waves() { return MODEL == sympa::MODEL_UPPER ? 1 : 2; }

template <int MODEL, int M, bool SCATTER>
__global__ __launch_bounds__(64, coop_bwd_waves<MODEL>()) void siegel_coop_bwd_kernel(const BwdArgs a, const int rounds) {
    constexpr bool UPPER = (MODEL == sympa::MODEL_UPPER);
    using namespace siegel_coop;
    using spd_coop::cholesky_rows;
    using spd_coop::solve_right_l;
    using spd_coop::solve_right_lt;
    using spd_coop::transpose_rows;
    __shared__ __attribute__((aligned(16))) double tbuf_all[spd_coop::GROUPS_PER_WAVE * spd_coop::TBUF];
    const DistArgs& f = a.f;
    const int lane = threadIdx.x;
    const int g = lane / spd_coop::GROUP, r = lane % spd_coop::GROUP;
    double* const tbuf = tbuf_all + g * spd_coop::TBUF;
    constexpr int nn = M * M;
    constexpr int64_t ROW = 2 * nn;
    double sc = 1.0;
    bool sc_active = false;
    if (f.scale != nullptr) {
        const double raw = f.scale[0] * f.inv_scale_coef;
        sc_active = raw > 0.1;
        sc = sc_active ? raw : 0.1;
    }
    int st = 0, nflag = 0;
    double loss_acc = 0.0, gscale_acc = 0.0;
    double gw_acc[M];
#pragma unroll
    for (int k = 0; k < M; ++k) gw_acc[k] = 0.0;
    // the training graph's batch window (DistArgs::batch_counter): pairs [c b, (c + 1) b) of the index / graph-distance lists
    const int64_t boff = (f.batch_counter != nullptr) ? f.batch_counter[0] * f.b : 0;

    for (int t = 0; t < rounds; ++t) {
        const int64_t first = ((int64_t)blockIdx.x * rounds + t) * spd_coop::GROUPS_PER_WAVE;
        if (first >= f.b) break;                                      // wave-uniform
        const int64_t i = first + g;                                  // my group's pair in this round
        const bool live = i < f.b;
        const int64_t ii = live ? i : f.b - 1;
        int64_t r1 = ii, r2 = ii;
        bool bad = false;
        if (f.idx1 != nullptr) {
            r1 = f.idx1[(ii + boff) * f.idx1_stride];
            r2 = f.idx2[(ii + boff) * f.idx2_stride];
            if (r1 < 0 || r1 >= f.num_rows || r2 < 0 || r2 >= f.num_rows) { bad = true; r1 = 0; r2 = 0; }
        }
        const double* pa = f.base1 + r1 * ROW;
        const double* pb = f.base2 + r2 * ROW;
        // upper: l1, l2 = rows of the real factors of Y1, Y2 (c1i, c2i unused);  bounded: c = rows of the complex factors
        double l1[M], l2[M], c1i[M], c2i[M], rd1[M], rd2[M];
        double etr[M], eti[M];
        bool pd1, pd2;
        if constexpr (UPPER) {
            double dr[M], di[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const int lo = r < j ? r : j, hi = r < j ? j : r;
                const int e = (hi < M) ? lo * M + hi : 0;             // upper triangle; a phantom lane reads element 0
                const double xa = pa[e], ya = pa[nn + e], xb = pb[e], yb = pb[nn + e];
                dr[j] = xb - xa; di[j] = yb - ya; l1[j] = ya; l2[j] = yb;
            }
            spd_coop::cholesky_rows2(l1, rd1, l2, rd2, pd1, pd2);
            spd_coop::solve_right_lt2(dr, di, l2, rd2);               // W = D L2^-T (both planes)
            transpose_rows(dr, etr, tbuf, r);
            transpose_rows(di, eti, tbuf, r);
            spd_coop::solve_right_lt2(etr, eti, l1, rd1);             // rows of E^T = W^T L1^-T: my column of E
        } else {
            double dr[M], di[M];
            {
                double w1r[M], w1i[M], w2r[M], w2i[M];
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int lo = r < j ? r : j, hi = r < j ? j : r;
                    const int e = (hi < M) ? lo * M + hi : 0;
                    w1r[j] = pa[e]; w1i[j] = pa[nn + e]; w2r[j] = pb[e]; w2i[j] = pb[nn + e];
                    dr[j] = w2r[j] - w1r[j]; di[j] = w2i[j] - w1i[j];
                }
                id_minus_wwh_rows(w2r, w2i, l2, c2i, r);              // A2 = I - W2 W2^H
                id_minus_wwh_rows(w1r, w1i, l1, c1i, r);
            }
            pd2 = ccholesky_rows(l2, c2i, rd2);
            csolve_right_lt(dr, di, l2, c2i, rd2);                    // W = D C2^-T
            transpose_rows(dr, etr, tbuf, r);
            transpose_rows(di, eti, tbuf, r);
            pd1 = ccholesky_rows(l1, c1i, rd1);
            csolve_right_lt(etr, eti, l1, c1i, rd1);                  // rows of E^T = W^T C1^-T
        }
        double d[M], e[M], vr[M], vi[M], bk[M], phr[M], phi_[M];
        {
            double hr[M], hi[M], br[M], bi[M];
            gram_columns(etr, eti, hr, hi);
            ctridiagonalize_keep(hr, hi, r, d, br, bi, vr, vi, bk);
            // T' = Phi T Phi^H with T real: e_k = |b_k|, Phi_{k+1} = Phi_k b_k / |b_k|
            phr[0] = 1.0; phi_[0] = 0.0;
            sfor<0, M - 1>([&](auto K) {
                constexpr int k = K;
                const double ab2 = sympa::d_fma(br[k], br[k], bi[k] * bi[k]);
                const double iab = sympa::d_rsqrt(ab2 + sympa::TINY);
                const bool zero = !(ab2 > 0.0);
                e[k] = ab2 * iab;
                const double ur = zero ? 1.0 : br[k] * iab, ui = zero ? 0.0 : bi[k] * iab;
                phr[k + 1] = sympa::d_fma(phr[k], ur, -phi_[k] * ui);
                phi_[k + 1] = sympa::d_fma(phr[k], ui, phi_[k] * ur);
            });
            e[M - 1] = 0.0;
        }
        // the QL below runs redundantly in the sixteen lanes of the pair and its predicates must agree bit for bit
#pragma unroll
        for (int j = 0; j < M; ++j) { d[j] = bcast<0>(d[j]); e[j] = bcast<0>(e[j]); }
        double zr[M], zi[M];
        bool conv;
        {
            double zrow[M];
#pragma unroll
            for (int j = 0; j < M; ++j) zrow[j] = (r == j) ? 1.0 : 0.0;
            conv = spd_coop::tridiag_ql_vectors_row(d, e, zrow);
            double myr = 0.0, myi = 0.0;                               // Phi_r of my row
#pragma unroll
            for (int k = 0; k < M; ++k) { myr = (r == k) ? phr[k] : myr; myi = (r == k) ? phi_[k] : myi; }
#pragma unroll
            for (int j = 0; j < M; ++j) { zr[j] = myr * zrow[j]; zi[j] = myi * zrow[j]; }
        }
        double vcr[M], vci[M];
        transpose_rows(zr, vcr, tbuf, r);                             // lane c holds column c of Phi Z
        transpose_rows(zi, vci, tbuf, r);
        cback_transform_columns(vcr, vci, vr, vi, bk);                // ... of V = Q Phi Z
        double utr[M], uti[M];
        ut_from_columns(etr, eti, vcr, vci, utr, uti);                // my column of U = E V
        double lam[M];
        {
            double l0 = 0.0, l1_ = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) { l0 = sympa::d_fma(utr[j], utr[j], l0); l1_ = sympa::d_fma(uti[j], uti[j], l1_); }
            const double mine = settle(l0 + l1_);                     // lambda_c = ||U[:, c]||^2, c = my lane
            sfor<0, M>([&](auto C) { lam[C] = bcast<C>(mine); });
        }
        double vrr[M], vri[M], urr[M], uri[M];
        transpose_rows(vcr, vrr, tbuf, r);                            // rows of V and of U
        transpose_rows(vci, vri, tbuf, r);
        transpose_rows(utr, urr, tbuf, r);
        transpose_rows(uti, uri, tbuf, r);

        // scalar part (group-uniform): metric value, spectral weights with go = 1, then the loss gives go
        double phi[M], philam[M], gwl[M];
#pragma unroll
        for (int k = 0; k < M; ++k) gwl[k] = 0.0;
        bool finite;
        double dist = sympa::spectral_adjoint<M, MODEL>(lam, f.metric, f.metric_w, f.inv_eps, 1.0, phi, philam,
                                                                     gwl, finite);
        if (!finite) dist = __builtin_nan("");
        double go = 0.0, loss_i = 0.0;
        if (a.graph_dist != nullptr) {   // AverageDistortionLoss (losses.py:10-19): sum |(d/g)^2 - 1|
            const double gd = live ? a.graph_dist[i + boff] : 1.0;
            const double ratio = dist * sc / gd;
            const double ee = ratio * ratio - 1.0;
            loss_i = (live && !bad) ? fabs(ee) * a.loss_scale : 0.0;
            go = (ee > 0.0 ? 1.0 : (ee < 0.0 ? -1.0 : 0.0)) * 2.0 * ratio / gd * a.loss_scale;
        } else {
            go = live ? a.go[i] : 0.0;
        }
        if (!live) go = 0.0;
        const double fs = go * sc;
        double p2[M];
#pragma unroll
        for (int k = 0; k < M; ++k) { p2[k] = 2.0 * fs * phi[k]; phi[k] *= fs; philam[k] *= fs; }

        // (g1r, g1i), (g2r, g2i): my rows of the symmetric gradients with respect to the two points
        double g1r[M], g1i[M], g2r[M], g2i[M];
        {
            double ebr[M], ebi[M];
            adbh_rows<M, false>(urr, uri, p2, vrr, vri, ebr, ebi);     // Ebar = 2 U diag(phi) V^H
            double tr[M], ti[M];
            if constexpr (UPPER) {
                // Dbar = L1^-T Ebar L2^-1, both planes; its symmetric part
#pragma unroll
                for (int j = 0; j < M; ++j) { l1[j] = settle(l1[j]); l2[j] = settle(l2[j]); }
                spd_coop::solve_right_l2(ebr, ebi, l2, rd2);
                transpose_rows(ebr, tr, tbuf, r);
                transpose_rows(ebi, ti, tbuf, r);
                spd_coop::solve_right_l2(tr, ti, l1, rd1);            // rows of Dbar^T
            } else {
                // Dbar = C1^-H Ebar conj(C2)^-1:  (C1^-H X)^T = X^T conj(C1)^-1
#pragma unroll
                for (int j = 0; j < M; ++j) { l1[j] = settle(l1[j]); l2[j] = settle(l2[j]); c1i[j] = settle(c1i[j]); c2i[j] = settle(c2i[j]); }
                csolve_right_l<M, true>(ebr, ebi, l2, c2i, rd2);
                transpose_rows(ebr, tr, tbuf, r);
                transpose_rows(ebi, ti, tbuf, r);
                csolve_right_l<M, true>(tr, ti, l1, c1i, rd1);        // rows of Dbar^T
            }
            transpose_rows(tr, ebr, tbuf, r);                         // rows of Dbar
            transpose_rows(ti, ebi, tbuf, r);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                g2r[j] = 0.5 * (tr[j] + ebr[j]); g2i[j] = 0.5 * (ti[j] + ebi[j]);
                g1r[j] = -g2r[j]; g1i[j] = -g2i[j];
            }
        }
        if constexpr (UPPER) {
            double gg[M], kq[M], dummy[M];
            adbh_rows<M, true>(urr, uri, phi, urr, uri, gg, dummy);    // Re G = Re U diag(phi) U^H
            adbh_rows<M, true>(vrr, vri, philam, vrr, vri, kq, dummy); // Re K = Re V diag(phi lambda) V^H
            spd_coop::congruence_inv_t_rows(gg, l1, rd1, tbuf, r);    // L1^-T Re G L1^-1
            spd_coop::congruence_inv_t_rows(kq, l2, rd2, tbuf, r);    // L2^-T Re K L2^-1
#pragma unroll
            for (int j = 0; j < M; ++j) { g1i[j] -= gg[j]; g2i[j] -= kq[j]; }
        } else {
            // Wbar_k = -/+ Dbar - 2 Abar_k W_k with Abar_1 = -C1^-H G C1^-1, Abar_2 = -C2^-H conj(K) C2^-1; symmetrised
            auto point_term = [&](double (&ar)[M], double (&ai)[M], double (&cr)[M], double (&ci)[M], const double (&rd)[M],
                                  const double* __restrict__ pw, double (&outr)[M], double (&outi)[M]) {
                ccongruence_inv_h_rows(ar, ai, cr, ci, rd, tbuf, r);  // C^-H (.) C^-1 = -Abar
                double wr[M], wi[M], tr[M], ti[M], sr[M], si[M];
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int lo = r < j ? r : j, hi = r < j ? j : r;
                    const int e = (hi < M) ? lo * M + hi : 0;
                    wr[j] = pw[e]; wi[j] = pw[nn + e];
                }
                cmatmul_rows(ar, ai, wr, wi, tr, ti);                 // T = (-Abar) W
                transpose_rows(tr, sr, tbuf, r);
                transpose_rows(ti, si, tbuf, r);
#pragma unroll
                for (int j = 0; j < M; ++j) { outr[j] += tr[j] + sr[j]; outi[j] += ti[j] + si[j]; }   // sym(-2 Abar W) = T + T^T
            };
            {
                double gr[M], gi[M];
                adbh_rows<M, false>(urr, uri, phi, urr, uri, gr, gi);  // G = U diag(phi) U^H
                point_term(gr, gi, l1, c1i, rd1, pa, g1r, g1i);
            }
            {
                double kr[M], ki[M];
                adbh_rows<M, false>(vrr, vri, philam, vrr, vri, kr, ki);
#pragma unroll
                for (int j = 0; j < M; ++j) ki[j] = -ki[j];           // conj(K)
                point_term(kr, ki, l2, c2i, rd2, pb, g2r, g2i);
            }
        }

        const bool write = live && !bad && r < M;
        if constexpr (SCATTER) {
            // One plane at a time through the group's LDS tile, so that consecutive lanes add to consecutive doubles: the
            // cost of an fp64 atomic wave-instruction is per 128-byte line it touches (profiles/r01_atomic_scope.txt), and
            // a lane-per-row instruction touches 64 lines where this one touches 4 (n = 16 fused step 1.13 -> see DESIGN 8).
            const bool on = live && !bad;
            double* o1 = a.g1 + r1 * ROW;
            double* o2 = a.g2 + r2 * ROW;
            spd_coop::scatter_plane<M>(g1r, tbuf, o1, r, on);
            spd_coop::scatter_plane<M>(g1i, tbuf, o1 + nn, r, on);
            spd_coop::scatter_plane<M>(g2r, tbuf, o2, r, on);
            spd_coop::scatter_plane<M>(g2i, tbuf, o2 + nn, r, on);
            (void)write;
        } else if (live && r < M) {
            double* o1 = a.g1 + i * ROW + r * M;
            double* o2 = a.g2 + i * ROW + r * M;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                o1[j] = bad ? 0.0 : g1r[j];
                o1[nn + j] = bad ? 0.0 : g1i[j];
                o2[j] = bad ? 0.0 : g2r[j];
                o2[nn + j] = bad ? 0.0 : g2i[j];
            }
        }
        if (r == 0) {
            if (live && f.out != nullptr) f.out[i] = bad ? __builtin_nan("") : dist * sc;
            loss_acc += loss_i;
            gscale_acc += (live && !bad && sc_active) ? go * dist * f.inv_scale_coef : 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) gw_acc[k] += (live && !bad) ? gwl[k] * fs : 0.0;
            if (live) {
                int s = 0;
                if (bad) s |= sympa::ST_BAD_INDEX;
                if (!(pd1 && pd2)) s |= sympa::ST_NOT_PD;
                if (!conv) s |= sympa::ST_NO_CONVERGENCE;
                if (!sympa::d_finite(dist)) s |= sympa::ST_NONFINITE;
                st |= s;
                nflag += (s != 0) ? 1 : 0;
            }
        }
    }
    // lanes r = 0 of the four groups hold the partial sums
    auto wave_sum = [&](double v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        return v;
    };
    if (a.gw != nullptr && f.metric == sympa::METRIC_WSUM) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
            const double x = wave_sum((r == 0) ? gw_acc[k] : 0.0);
            if (lane == 0 && x != 0.0) atomicAdd(a.gw + k, x);
        }
    }
    if (a.gscale != nullptr && f.scale != nullptr) {
        const double x = wave_sum((r == 0) ? gscale_acc : 0.0);
        if (lane == 0 && x != 0.0) atomicAdd(a.gscale, x);
    }
    if (a.loss != nullptr && a.graph_dist != nullptr) {
        const double x = wave_sum((r == 0) ? loss_acc : 0.0);
        if (lane == 0 && x != 0.0) atomicAdd(a.loss, x);
    }
    if (f.status != nullptr) {
        if (__ballot(st != 0) != 0ull) {
            if (st != 0) atomicOr(&f.status[0], st);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nflag += __shfl_xor(nflag, off);
            if (lane == 0) atomicAdd(&f.status[1], nflag);
        }
    }
}

template <int MODEL, int M, bool SCATTER>
int launch_coop_bwd_ms(const BwdArgs& a, hipStream_t s) {
    constexpr int PPR = spd_coop::GROUPS_PER_WAVE;               // pairs per wave and round
    const int rounds = spd_coop::coop_rounds(a.f.b, coop_bwd_waves<MODEL>(), PPR);
    const dim3 grid((unsigned)((a.f.b + PPR * rounds - 1) / (PPR * rounds)));
    hipLaunchKernelGGL((siegel_coop_bwd_kernel<MODEL, M, SCATTER>), grid, dim3(64), 0, s, a, rounds);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

// n = 9..16, both models (siegel_bwd_coop_<model>_<n>_{dense,scatter}.hip: one kernel per translation unit; siegel_bwd_coop.hip)
int launch_bwd_coop(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s);
int launch_bwd_coop_upper_9_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_9_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_10_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_10_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_11_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_11_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_12_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_12_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_13_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_13_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_14_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_14_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_15_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_15_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_16_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_upper_16_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_9_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_9_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_10_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_10_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_11_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_11_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_12_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_12_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_13_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_13_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_14_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_14_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_15_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_15_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_16_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_coop_bounded_16_scatter(const BwdArgs& a, hipStream_t s);

}  // namespace sympa_hip
