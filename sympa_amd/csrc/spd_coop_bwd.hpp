// SPD model, backward with sixteen lanes per pair (the layout and DPP machinery of spd_coop.hpp): the training path of
// BASELINE.json configs[4] (spd, n = 16, batch 1 M).  The one-lane-per-pair backward (spd_math_bwd.hpp) keeps five
// 16 x 16 matrices per lane in scratch memory and runs at ~5 M pairs/s; here everything of a pair is spread over the
// sixteen lanes of a DPP row, one matrix ROW per lane, and stays in registers:
//
//   Cholesky x = L L^T, W = (y - x) L^-T, transpose, A = W^T L^-T                      (the forward's first half)
//   Householder tridiagonalisation A = Q T Q^T with the reflectors KEPT: lane r keeps component r of every v_k,
//       beta_k and the signed off-diagonals are group-uniform
//   implicit QL with the rotations accumulated (tql2): the scalar recurrence on (d, e) is group-uniform -- every lane
//       of the group runs it redundantly -- and a rotation of columns i, i+1 of Z is four instructions on the lane's
//       own row of Z; per-group active block [L, m] by predicates, iteration ended when all four pairs of the wave agree
//   back-transformation V = P_0 ... P_{M-3} Z in COLUMN layout (transpose through LDS): lane c holds column c, so
//       tau_c = beta v^T z_c and z_c -= tau_c v are two fmac_dpp per element with v broadcast from the lane that owns
//       the component -- 210 instructions for all reflectors instead of a 16-lane reduction per column
//   P = V diag(g) V^T row by row, then L^-T P L^-1 through `p <- p L^-1`, transpose, `p <- p L^-1` (P symmetric)
//   d dist / d y: g_i = log(1 + a_i) / ((1 + a_i) dist);   d dist / d x: g_i = -log(1 + a_i) / dist
//
// Same formulas as spd_math_bwd.hpp; the tests check the two kernels against each other, against the 50-digit
// finite differences and against autograd through the oracle.
#pragma once

#include "spd_coop.hpp"

namespace spd_coop {

// a <- a L^-1 for the rows held one per lane:  a'[j] = (a[j] - sum_{k>j} a'[k] L[k][j]) / L[j][j];  L[k][j] is register j
// of lane k.
template <int M>
__device__ __forceinline__ void solve_right_l(double (&a)[M], const double (&l)[M], const double (&rd)[M]) {
    sfor<0, M>([&](auto JJ) {
        constexpr int j = M - 1 - JJ;
        sfor<j + 1, M>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<k>(a[j], l[j], a[k]);
        });
        a[j] *= rd[j];
    });
}

// two right-hand sides at once (independent chains interleaved in the source: the DPP statements keep program order)
template <int M>
__device__ __forceinline__ void solve_right_l2(double (&a)[M], double (&b)[M], const double (&l)[M], const double (&rd)[M]) {
    sfor<0, M>([&](auto JJ) {
        constexpr int j = M - 1 - JJ;
        sfor<j + 1, M>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<k>(a[j], l[j], a[k]);
            fnmac_bc<k>(b[j], l[j], b[k]);
        });
        a[j] *= rd[j];
        b[j] *= rd[j];
    });
}

// Householder tridiagonalisation of the symmetric matrix held one row per lane, reflectors kept:
//   d[k], e[k] (signed: e[k] = T[k+1][k]) group-uniform;  vk[k] = my component of reflector k;  bk[k] = beta_k.
template <int M>
__device__ __forceinline__ void tridiagonalize_keep(double (&m)[M], const int r, double (&d)[M], double (&e)[M],
                                                    double (&vk)[M], double (&bk)[M]) {
    sfor<0, M - 2>([&](auto K) {
        constexpr int k = K;
        const double col = settle(m[k]);
        const double x0 = bcast<k + 1>(col);
        const double dk = bcast<k>(col);
        const double tail = (r > k + 1 && r < M) ? col : 0.0;
        const double s2 = group_sum(tail * tail);
        const double n2 = sympa::d_fma(x0, x0, s2);
        const double nx = sympa::d_sqrt(n2);
        const double sx = copysign(nx, x0);
        d[k] = dk;
        e[k] = -sx;                                  // P x = -sign(x0) ||x|| e1
        const double v0 = x0 + sx;
        const double den = sympa::d_fma(v0, v0, s2);
        const double beta = (den > 0.0) ? 2.0 * sympa::d_rcp(den) : 0.0;
        const double vi = settle((r <= k || r >= M) ? 0.0 : ((r == k + 1) ? v0 : col));
        vk[k] = vi;
        bk[k] = beta;
        double ps[4] = {0.0, 0.0, 0.0, 0.0};
        sfor<k + 1, M>([&](auto J) { fmac_bc<J>(ps[J % 4], vi, m[J]); });
        double p = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        p = (r <= k || r >= M) ? 0.0 : beta * p;
        const double kk = 0.5 * beta * group_sum(vi * p);
        const double q = settle(sympa::d_fma(-kk, vi, p));
        sfor<k + 1, M>([&](auto J) {
            constexpr int j = J;
            fnmac_bc<j>(m[j], vi, q);
            fnmac_bc<j>(m[j], q, vi);
        });
    });
    const double last = settle(m[M - 1]);
    d[M - 2] = bcast<M - 2>(settle(m[M - 2]));
    d[M - 1] = bcast<M - 1>(last);
    e[M - 2] = bcast<M - 2>(last);                   // T[M-1][M-2] = A[M-2][M-1]
    e[M - 1] = 0.0;
    vk[M - 2] = 0.0; vk[M - 1] = 0.0; bk[M - 2] = 0.0; bk[M - 1] = 0.0;
}

// Implicit QL with accumulated rotations (tql2) on the group-uniform tridiagonal (d, e); zrow = my row of Z (identity
// on entry).  The iteration of a stage ends when all pairs of the wave have deflated position L.  The per-pair active
// block [L, m] is handled by SELECTS on every state variable (18 of the ~49 instructions of a step).  Real branches --
// the sixteen lanes of a pair take them together, so the hardware's exec mask would do the predication -- were tried:
// 6-12 % faster (spd n = 16 1.28 -> 1.21 ms, upper n = 16 2.83 -> 2.49 ms per 65 536 pairs) and correct in every kernel
// but one (upper, n = 16, scatter output: every gradient wrong with the distances right, identical with and without wait
// states in the asm statements, i.e. not a hazard; the same source was right in the rows-out kernel).  Not understood,
// so not used.
template <int M>
__device__ __forceinline__ bool tridiag_ql_vectors_row(double (&d)[M], double (&e)[M], double (&zrow)[M]) {
    bool all_ok = true;
    sfor<0, M - 1>([&](auto LL) {
        constexpr int L = LL;
        bool conv = false;
        for (int it = 0; it < 50; ++it) {
            int mm = M - 1;
            sfor<0, M - 1 - L>([&](auto II) {
                constexpr int i = M - 2 - II;
                const bool negl = sympa::ql_negligible(e[i] * e[i], d[i], d[i + 1]);
                e[i] = negl ? 0.0 : e[i];
                mm = negl ? i : mm;
            });
            conv = (mm == L);
            if (sympa::wave_all(conv)) break;
            const double el = conv ? 1.0 : e[L];
            const double g0 = 0.5 * (d[L + 1] - d[L]) * sympa::d_rcp(el);
            const double r0 = sympa::d_sqrt(sympa::d_fma(g0, g0, 1.0));
            const double shift = el * sympa::d_rcp(g0 + copysign(r0, g0)) - d[L];
            double c = 1.0, s = 1.0, p = 0.0, g = 0.0;
            sfor<0, M - 1 - L>([&](auto II) {
                constexpr int i = M - 2 - II;
                const bool active = !conv && (i < mm);
                const bool start = (mm == i + 1);
                g = start ? d[i + 1] + shift : g;
                s = start ? 1.0 : s;
                c = start ? 1.0 : c;
                p = start ? 0.0 : p;
                const double f = s * e[i];
                const double b = c * e[i];
                const double r2 = sympa::d_fma(f, f, g * g);
                const double ir = sympa::d_rsqrt(r2 + sympa::TINY);
                const double rr0 = r2 * ir;
                const bool rzero = !(r2 > 0.0);
                const double sn = rzero ? 0.0 : f * ir;
                const double cn = rzero ? 1.0 : g * ir;
                if constexpr (i + 1 <= M - 2) e[i + 1] = (active && !start) ? rr0 : e[i + 1];
                const double g2 = d[i + 1] - p;
                const double rr = sympa::d_fma(d[i] - g2, sn, 2.0 * cn * b);
                const double pn = sn * rr;
                d[i + 1] = active ? g2 + pn : d[i + 1];
                const double gn = sympa::d_fma(cn, rr, -b);
                const double ce = active ? cn : 1.0, se = active ? sn : 0.0;
                const double zf = zrow[i + 1];
                zrow[i + 1] = sympa::d_fma(se, zrow[i], ce * zf);
                zrow[i] = sympa::d_fma(ce, zrow[i], -se * zf);
                s = active ? sn : s;
                c = active ? cn : c;
                p = active ? pn : p;
                g = active ? gn : g;
            });
            d[L] = conv ? d[L] : d[L] - p;
            e[L] = conv ? e[L] : g;
        }
        all_ok = all_ok && conv;
    });
    return all_ok;
}

// The same iteration with TWO rows of Z per lane: eight lanes then carry the sixteen rows of a pair, so the two halves of
// a group of sixteen run the QL of two different pairs and every instruction of the scalar recurrence (two thirds of the
// whole backward kernel) serves eight pairs per wave instead of four.  (d, e) must be uniform over the eight lanes.
template <int M>
__device__ __forceinline__ bool tridiag_ql_vectors_2rows(double (&d)[M], double (&e)[M], double (&zrow)[M], double (&zrw2)[M]) {
    bool all_ok = true;
    sfor<0, M - 1>([&](auto LL) {
        constexpr int L = LL;
        bool conv = false;
        for (int it = 0; it < 50; ++it) {
            int mm = M - 1;
            sfor<0, M - 1 - L>([&](auto II) {
                constexpr int i = M - 2 - II;
                const bool negl = sympa::ql_negligible(e[i] * e[i], d[i], d[i + 1]);
                e[i] = negl ? 0.0 : e[i];
                mm = negl ? i : mm;
            });
            conv = (mm == L);
            if (sympa::wave_all(conv)) break;
            const double el = conv ? 1.0 : e[L];
            const double g0 = 0.5 * (d[L + 1] - d[L]) * sympa::d_rcp(el);
            const double r0 = sympa::d_sqrt(sympa::d_fma(g0, g0, 1.0));
            const double shift = el * sympa::d_rcp(g0 + copysign(r0, g0)) - d[L];
            double c = 1.0, s = 1.0, p = 0.0, g = 0.0;
            sfor<0, M - 1 - L>([&](auto II) {
                constexpr int i = M - 2 - II;
                const bool active = !conv && (i < mm);
                const bool start = (mm == i + 1);
                g = start ? d[i + 1] + shift : g;
                s = start ? 1.0 : s;
                c = start ? 1.0 : c;
                p = start ? 0.0 : p;
                const double f = s * e[i];
                const double b = c * e[i];
                const double r2 = sympa::d_fma(f, f, g * g);
                const double ir = sympa::d_rsqrt(r2 + sympa::TINY);
                const double rr0 = r2 * ir;
                const bool rzero = !(r2 > 0.0);
                const double sn = rzero ? 0.0 : f * ir;
                const double cn = rzero ? 1.0 : g * ir;
                if constexpr (i + 1 <= M - 2) e[i + 1] = (active && !start) ? rr0 : e[i + 1];
                const double g2 = d[i + 1] - p;
                const double rr = sympa::d_fma(d[i] - g2, sn, 2.0 * cn * b);
                const double pn = sn * rr;
                d[i + 1] = active ? g2 + pn : d[i + 1];
                const double gn = sympa::d_fma(cn, rr, -b);
                const double ce = active ? cn : 1.0, se = active ? sn : 0.0;
                const double zf = zrow[i + 1], zg = zrw2[i + 1];
                zrow[i + 1] = sympa::d_fma(se, zrow[i], ce * zf);
                zrow[i] = sympa::d_fma(ce, zrow[i], -se * zf);
                zrw2[i + 1] = sympa::d_fma(se, zrw2[i], ce * zg);
                zrw2[i] = sympa::d_fma(ce, zrw2[i], -se * zg);
                s = active ? sn : s;
                c = active ? cn : c;
                p = active ? pn : p;
                g = active ? gn : g;
            });
            d[L] = conv ? d[L] : d[L] - p;
            e[L] = conv ? e[L] : g;
        }
        all_ok = all_ok && conv;
    });
    return all_ok;
}

// zc = my COLUMN of Z (lane c holds Z[:, c]);  zc <- P_0 ... P_{M-3} zc  with P_k = I - beta_k v_k v_k^T, component
// r of v_k living in lane r's vk[k].
template <int M>
__device__ __forceinline__ void back_transform_columns(double (&zc)[M], double (&vk)[M], const double (&bk)[M]) {
    sfor<0, M - 2>([&](auto KK) {
        constexpr int k = M - 3 - KK;
        const double vsrc = settle(vk[k]);
        double t0 = 0.0, t1 = 0.0;
        sfor<k + 1, M>([&](auto R) {
            constexpr int rr = R;
            if constexpr (rr % 2 == 0) fmac_bc<rr>(t0, vsrc, zc[rr]);
            else fmac_bc<rr>(t1, vsrc, zc[rr]);
        });
        const double tau = bk[k] * (t0 + t1);
        sfor<k + 1, M>([&](auto R) {
            constexpr int rr = R;
            fnmac_bc<rr>(zc[rr], vsrc, tau);
        });
    });
}

// prow[j] = sum_c V[me][c] g_c V[j][c]  (row `me` of V diag(g) V^T) from the rows of V held one per lane; g uniform.
template <int M>
__device__ __forceinline__ void vdvt_rows(double (&vrow)[M], const double (&g)[M], double (&prow)[M]) {
    double sc[M];
    sfor<0, M>([&](auto C) { vrow[C] = settle(vrow[C]); sc[C] = vrow[C] * g[C]; });
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        double a0 = 0.0, a1 = 0.0;
        sfor<0, M>([&](auto C) {
            constexpr int c = C;
            if constexpr (c % 2 == 0) fmac_bc<j>(a0, vrow[c], sc[c]);
            else fmac_bc<j>(a1, vrow[c], sc[c]);
        });
        prow[j] = a0 + a1;
    });
}

// p <- L^-T p L^-1 for a symmetric p held one row per lane (l = rows of L, rd = 1 / diag)
template <int M>
__device__ __forceinline__ void congruence_inv_t_rows(double (&p)[M], double (&l)[M], const double (&rd)[M],
                                                      double* __restrict__ tbuf, const int r) {
    sfor<0, M>([&](auto J) { l[J] = settle(l[J]); });
    solve_right_l(p, l, rd);                 // p L^-1
    double t[M];
    transpose_rows(p, t, tbuf, r);           // (p L^-1)^T = L^-T p   (p symmetric)
    solve_right_l(t, l, rd);                 // L^-T p L^-1
    sfor<0, M>([&](auto J) { p[J] = t[J]; });
}

}  // namespace spd_coop
