// The Siegel adjoint (siegel_math_bwd.hpp) cut in two at its narrowest point, for dims 5..8, ONE PAIR PER LANE.
//
// Why: the whole adjoint of an 8 x 8 pair keeps E, V, both factors and two or three products alive at once (~600 doubles
// against 256 per lane), so the one-kernel form spills ~1100 registers to scratch, and the lanes-per-pair form that avoids the
// spills (siegel_coop_bwd_kernel.hpp, eight lanes per pair) runs every scalar recurrence -- the QL above all -- redundantly
// in the eight lanes of a pair and issues every cross-lane FMA twice (profiles/r04_n8_backward_split.txt: front 22 %, QL 34 %,
// the products and solves behind it 43 % of 1.51 ms per 262 144 pairs).  With
//     Hbar = V diag(phi) V^H   (Hermitian n x n)    and    K = V diag(phi lambda) V^H = Hbar H
// everything behind the eigen-decomposition is a function of E, the factors and these two matrices:
//     Ebar = 2 E Hbar,   G = E Hbar E^H = Ebar E^H / 2,   Dbar = L1^-H Ebar conj(L2)^-1,   A1bar = -L1^-H G L1^-1,
//     A2bar = -L2^-H conj(K) L2^-1
// so the adjoint is two stages that share nothing but `AdjPack` (100 doubles per pair at n = 8, upper model):
//   stage 1 (pair_adjoint_spectral): factors, E, H = E^H E, eigen-decomposition WITH vectors (Householder + QL, one pair per
//            lane: 64 pairs per instruction stream), metric value, spectral weights with go = 1, Hbar and K.  E and the
//            factors are dead before the eigenvectors exist.
//   stage 2 (pair_adjoint_gradient): factors and E AGAIN from the table rows (cheap one pair per lane: ~2.6 k instructions),
//            then the products, solves and congruences above, scaled by go * scale.
// The two stages are separate kernels (siegel_bwd_split_kernel.hpp) with the packs in a caller-owned workspace laid out
// [entry][pair]: every load and store of the workspace is a contiguous 512 bytes per wave.
//
// Difference from pair_backward: the eigenvalues that enter phi are the QL's (accurate to eps ||H||), not the Rayleigh
// quotients ||E v_i||^2 -- E is gone when the vectors exist.  riem and finf do not see it (phi_i = 2 / (d (1 + lambda_i))): equal to
// rounding down to a spread of 1e-12.  fone / fmin / wsum (phi_i ~ lambda_i^-1/2) would carry eps lambda_max / lambda_i (rounds 4, 5:
// 5e-9 of the gradient at a spread of 1e-8, 7e-5 at 1e-12, profiles/r05_split_graded_spectrum.txt); since round 6 a wave that holds
// such a pair (spread < SPLIT_GRADED_RATIO) refines its eigenvalues to the same quotients -- see "Graded spectra" below.  The
// dispatcher keeps the one-stage kernels reachable (SYMPA_FLAG_GENERIC / no workspace); tests/test_backward_split.py compares both.
#pragma once

#include "siegel_math_bwd.hpp"
#include "tridiag_invit.hpp"

// SYMPA_PIN(x): the value x exists in a register at this point of the program order of side effects (device build); nothing on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define SYMPA_PIN(x) asm volatile("" ::"v"(x))
#else
#define SYMPA_PIN(x) ((void)(x))
#endif

namespace sympa {

template <int N, int MODEL>
struct AdjPack {
    static constexpr int OFFD = N * (N - 1) / 2;
    static constexpr int H_D = 0;                       // Hbar: diagonal [N]
    static constexpr int H_RE = N;                      //       strict upper triangle, row-major (j < k), real parts
    static constexpr int H_IM = N + OFFD;               //       ... imaginary parts
    static constexpr int K_D = N + 2 * OFFD;            // K:    diagonal
    static constexpr int K_RE = K_D + N;                //       strict upper, real
    static constexpr int K_IM = K_RE + OFFD;            //       strict upper, imaginary (bounded model only)
    static constexpr int LEN = K_RE + OFFD + (MODEL == MODEL_UPPER ? 0 : OFFD);
};
SYMPA_HD constexpr int offd_index(int n, int j, int k) { return j * n - j * (j + 1) / 2 + (k - j - 1); }   // j < k

// ---------------------------------------------------------------------------------------------
// Eigen-decomposition of the Hermitian h for stage 1, n >= 5: Householder form (reflectors kept), eigenvalues by the forward's
// lockstep eigenvalue-only QL (siegel_math.hpp: ~30 instructions per element-sweep, no per-lane block bookkeeping), eigenvectors of
// the real tridiagonal T by inverse iteration (tridiag_invit.hpp: O(n) per vector, modified Gram-Schmidt inside clusters of close
// eigenvalues), then V = Q Phi Z.  herm_eigen_vectors_ql's QL WITH accumulated rotations sweeps the whole static index range
// with selects and updates 2 n doubles of Z per element: ~11 k of the 28 k instructions of the spectral kernel at n = 8 against
// ~2.5 k + ~2.5 k here.  A pair with a cluster of more than INVIT_KEEP + 1 close eigenvalues (y = c x: every eigenvalue equal) sends
// its WHOLE WAVE through the QL with vectors instead (wave-uniform branch; a generic batch never takes it).
// MEASURED AND NOT ADOPTED (-DSYMPA_SPLIT_EIGEN_INVIT): fused step, upper n = 8, 262 144 pairs 818 -> 850 us.  The instruction
// count does drop, but both routes have to be in the kernel (24 k static instructions = 190 KB of straight-line code that every
// wave fetches once, against 17 k), and the inverse iteration's LU / solve / Gram-Schmidt temporaries next to the reflectors spill
// 832 bytes where the QL form spills 24.  Equal results (tests/test_backward_split.py runs both on the CPU build).
// Eigenvalues into h.d (ascending on the inverse-iteration route), eigenvectors into the columns of v.
// ---------------------------------------------------------------------------------------------
template <int N>
SYMPA_HD bool herm_eigen_vectors_invit(Herm<N>& h, CMat<N>& v) {
    double a[N], e[N], phr[N], phi[N], beta[N];
    CMat<N> refl;
    herm_tridiagonalize_keep<N>(h, a, e, phr, phi, refl, beta);
    double lam[N];
    bool ok;
    {
        double d[N], e2[N];
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) { d[i] = a[i]; e2[i] = e[i] * e[i]; }
        ok = tridiag_ql_lockstep<N>(d, e2);
        sort_ascending<N>(d);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) lam[i] = d[i];
    }
    double z[N][N];
    const bool small_blocks = tridiag_eigvecs_invit<N>(a, e, lam, [&](auto IC, const double (&x)[N]) {
        constexpr int c = decltype(IC)::value;
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) z[j][c] = x[j];
    });
    if (!wave_all(small_blocks)) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) z[i][j] = (i == j) ? 1.0 : 0.0;
        ok = tridiag_ql_vectors<N>(a, e, z);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) lam[i] = a[i];
    }
    // W = Phi Z,  Phi_0 = 1,  Phi_{j+1} = Phi_j * phase(T[j+1][j]):  T = Phi T_real Phi^H
    double fr = 1.0, fi = 0.0;
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) { v.re[j][i] = fr * z[j][i]; v.im[j][i] = fi * z[j][i]; }
        if (j < N - 1) {
            const double nr = fr * phr[j] - fi * phi[j];
            const double ni = fr * phi[j] + fi * phr[j];
            fr = nr; fi = ni;
        }
    }
    // V = P_0 P_1 ... P_{N-3} W,  P_k = I - beta_k v_k v_k^H on the rows k+1 .. N-1
SYMPA_UNROLL
    for (int k = N - 3; k >= 0; --k) {
SYMPA_UNROLL
        for (int c = 0; c < N; ++c) {
            double tr = 0.0, ti = 0.0;     // tau = beta * v^H w
SYMPA_UNROLL
            for (int i = k + 1; i < N; ++i) {
                tr = d_fma(refl.re[i][k], v.re[i][c], d_fma(refl.im[i][k], v.im[i][c], tr));
                ti = d_fma(refl.re[i][k], v.im[i][c], d_fma(-refl.im[i][k], v.re[i][c], ti));
            }
            tr *= beta[k]; ti *= beta[k];
SYMPA_UNROLL
            for (int i = k + 1; i < N; ++i) {
                v.re[i][c] = d_fma(-tr, refl.re[i][k], d_fma(ti, refl.im[i][k], v.re[i][c]));
                v.im[i][c] = d_fma(-tr, refl.im[i][k], d_fma(-ti, refl.re[i][k], v.im[i][c]));
            }
        }
    }
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) h.d[i] = lam[i];
    return ok;
}

// ---------------------------------------------------------------------------------------------
// Graded spectra (round 6).  The QL's eigenvalues of H = E^H E are accurate to eps ||H||; fone / fmin / wsum weight eigenvalue i with
// phi_i ~ lambda_i^-1/2, so a spread lambda_min / lambda_max = r leaves the relative error eps / r in the weights of the small ones
// (7e-5 of the gradient at r = 1e-12, profiles/r05_split_graded_spectrum.txt).  The one-stage adjoint refines every eigenvalue to the
// Rayleigh quotient ||E v_i||^2; stage 1 has lost E by the time the vectors exist, so it does the same ONLY where it matters: when a
// pair of the wave has r < SPLIT_GRADED_RATIO under one of those metrics, `refine(v, lambda)` is called (wave-uniform branch; a
// generic batch never takes it).  The default `refine` still has the points (CPU build, tests) and replaces the eigenvalues by the
// quotients with E formed AGAIN.  The gfx950 spectral kernel only takes note: the wave hands on zero packs and a flag, and a third
// launch runs the one-stage kernel (pair_backward: the same quotients) on the flagged waves (siegel_bwd_split_kernel.hpp,
// siegel_bwd_split.hip).  [Doing the refinement inside the spectral kernel -- V parked in the workspace, a not-inlined function
// that reloads the rows -- was built first: the call's spills landed next to the gather's hand-counted s_waitcnt vmcnt and the
// common path read its LDS tile early (wrong gradients at n = 7, +6 % time at n = 8).  Nothing of the rare path lives there now.]
// ---------------------------------------------------------------------------------------------
constexpr double SPLIT_GRADED_RATIO = 1e-5;       // eps / 1e-5 = 2e-11 of relative error in a weight is what the QL route may leave

SYMPA_HD bool metric_wants_relative_eigenvalues(const int metric) {
    return metric == METRIC_FONE || metric == METRIC_FMIN || metric == METRIC_WSUM;
}

// lambda_c = || E v_c ||^2 with E = L1^-1 (Z2 - Z1) L2^-T formed from the points; `column(c, vr, vi)` hands over eigenvector c
template <int N, int MODEL, class Column>
SYMPA_HD void rayleigh_quotients_from_points(const CMat<N>& z1, const CMat<N>& z2, Column&& column, double (&lam)[N]) {
    constexpr bool CPLX = (MODEL == MODEL_BOUNDED);
    CMat<N> e;
    {
        Tri<N, CPLX> l1, l2;
        if constexpr (MODEL == MODEL_UPPER) {
            (void)chol_real<N>(z1.im, l1);
            (void)chol_real<N>(z2.im, l2);
        } else {
            (void)chol_id_minus_wwh<N>(z1, l1);
            (void)chol_id_minus_wwh<N>(z2, l2);
        }
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) {
                e.re[i][j] = z2.re[i][j] - z1.re[i][j];
                e.im[i][j] = z2.im[i][j] - z1.im[i][j];
            }
        solve_left<N, CPLX>(l1, e);
        solve_right_t<N, CPLX>(l2, e);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int c = 0; c < N; ++c) {
        double vr[N], vi[N];
        column(c, vr, vi);
        double acc = 0.0;
SYMPA_UNROLL
        for (int r = 0; r < N; ++r) {
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) {
                tr = d_fma(e.re[r][k], vr[k], d_fma(-e.im[r][k], vi[k], tr));
                ti = d_fma(e.re[r][k], vi[k], d_fma(e.im[r][k], vr[k], ti));
            }
            acc = d_fma(tr, tr, d_fma(ti, ti, acc));
        }
        lam[c] = acc;
    }
}

// the default `refine` of pair_adjoint_spectral: the caller still holds the points
template <int N, int MODEL>
struct RefineFromPoints {
    const CMat<N>& z1;
    const CMat<N>& z2;
    SYMPA_HD void operator()(CMat<N>& v, double (&lam)[N]) const {
        // (column c by a run-time index: a copy of V the loop can index -- this is the CPU build's path)
        rayleigh_quotients_from_points<N, MODEL>(z1, z2, [&](const int c, double (&vr)[N], double (&vi)[N]) {
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) { vr[k] = v.re[k][c]; vi[k] = v.im[k][c]; }
        }, lam);
    }
};

// ---------------------------------------------------------------------------------------------
// Stage 1.  Returns the metric value (NaN for non-finite input); pack = Hbar, K for go = 1; gw[k] += d out / d w_k.
// ---------------------------------------------------------------------------------------------
template <int N, int MODEL, class Refine>
SYMPA_HD double pair_adjoint_spectral(const CMat<N>& z1, const CMat<N>& z2, int metric, const double* __restrict__ w,
                                      double inv_eps, double (&pack)[AdjPack<N, MODEL>::LEN], double (&gw)[N], int& status,
                                      Refine&& refine) {
    using P = AdjPack<N, MODEL>;
    constexpr bool CPLX = (MODEL == MODEL_BOUNDED);
    Herm<N> h;
    bool ok;
    {
        Tri<N, CPLX> l1, l2;
        CMat<N> e;
        if constexpr (MODEL == MODEL_UPPER) {
            ok = chol_real<N>(z1.im, l1);
            ok = chol_real<N>(z2.im, l2) && ok;
        } else {
            ok = chol_id_minus_wwh<N>(z1, l1);
            ok = chol_id_minus_wwh<N>(z2, l2) && ok;
        }
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) {
                e.re[i][j] = z2.re[i][j] - z1.re[i][j];
                e.im[i][j] = z2.im[i][j] - z1.im[i][j];
            }
        solve_left<N, CPLX>(l1, e);
        solve_right_t<N, CPLX>(l2, e);
        gram<N>(e, h);
    }
    CMat<N> v;
    bool conv;
    if constexpr (N >= 5) conv = herm_eigen_vectors_ql<N>(h, v);
    else conv = herm_eigen_vectors<N>(h, v);
    if constexpr (N >= 5) {
        bool graded = false;
        if (metric_wants_relative_eigenvalues(metric)) {
            double lo = h.d[0], hi = h.d[0];
SYMPA_UNROLL
            for (int i = 1; i < N; ++i) { lo = fmin(lo, h.d[i]); hi = fmax(hi, h.d[i]); }
            graded = lo < SPLIT_GRADED_RATIO * hi;          // (NaN eigenvalues: false)
        }
        if (!wave_all(!graded)) refine(v, h.d);
    }

    double phi[N], philam[N];
    bool finite;
    double out = spectral_adjoint<N, MODEL>(h.d, metric, w, inv_eps, 1.0, phi, philam, gw, finite);

    // Hbar_jk = sum_i phi_i V_ji conj(V_ki),  K_jk = sum_i phi_i lambda_i V_ji conj(V_ki)      (j <= k)
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
SYMPA_UNROLL
        for (int k = j; k < N; ++k) {
            double hr = 0.0, hi = 0.0, kr = 0.0, ki = 0.0;
SYMPA_UNROLL
            for (int i = 0; i < N; ++i) {
                const double pr = d_fma(v.re[j][i], v.re[k][i], v.im[j][i] * v.im[k][i]);     // Re V_ji conj(V_ki)
                hr = d_fma(phi[i], pr, hr);
                kr = d_fma(philam[i], pr, kr);
                if (k > j) {
                    const double pi = d_fma(v.im[j][i], v.re[k][i], -v.re[j][i] * v.im[k][i]);
                    hi = d_fma(phi[i], pi, hi);
                    if (CPLX) ki = d_fma(philam[i], pi, ki);
                }
            }
            if (k == j) {
                pack[P::H_D + j] = hr;
                pack[P::K_D + j] = kr;
            } else {
                pack[P::H_RE + offd_index(N, j, k)] = hr;
                pack[P::H_IM + offd_index(N, j, k)] = hi;
                pack[P::K_RE + offd_index(N, j, k)] = kr;
                if constexpr (CPLX) pack[P::K_IM + offd_index(N, j, k)] = ki;
            }
        }
    }
    if (!finite) out = __builtin_nan("");     // non-finite input: the forward value is NaN like the gradients
    if (!ok) status |= ST_NOT_PD;
    if (!conv) status |= ST_NO_CONVERGENCE;
    if (!d_finite(out)) status |= ST_NONFINITE;
    return out;
}

template <int N, int MODEL>
SYMPA_HD double pair_adjoint_spectral(const CMat<N>& z1, const CMat<N>& z2, int metric, const double* __restrict__ w,
                                      double inv_eps, double (&pack)[AdjPack<N, MODEL>::LEN], double (&gw)[N], int& status) {
    return pair_adjoint_spectral<N, MODEL>(z1, z2, metric, w, inv_eps, pack, gw, status, RefineFromPoints<N, MODEL>{z1, z2});
}

// ---------------------------------------------------------------------------------------------
// Stage 2.  pack: Hbar and K as written by stage 1 (any common factor -- go * scale -- may have been applied to all of it:
// the gradients are linear in the pack).  g1, g2: the symmetric matrix gradients.
// ---------------------------------------------------------------------------------------------
template <int N, int MODEL>
SYMPA_HD void pair_adjoint_gradient(const CMat<N>& z1, const CMat<N>& z2, const double (&pack)[AdjPack<N, MODEL>::LEN],
                                    CMat<N>& g1, CMat<N>& g2) {
    using P = AdjPack<N, MODEL>;
    constexpr bool CPLX = (MODEL == MODEL_BOUNDED);
    Tri<N, CPLX> l1, l2;
    CMat<N> e;
    if constexpr (MODEL == MODEL_UPPER) {
        (void)chol_real<N>(z1.im, l1);
        (void)chol_real<N>(z2.im, l2);
    } else {
        (void)chol_id_minus_wwh<N>(z1, l1);
        (void)chol_id_minus_wwh<N>(z2, l2);
    }
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            e.re[i][j] = z2.re[i][j] - z1.re[i][j];
            e.im[i][j] = z2.im[i][j] - z1.im[i][j];
        }
    solve_left<N, CPLX>(l1, e);
    solve_right_t<N, CPLX>(l2, e);

    CMat<N> ebar, a1, a2;
    {
        CMat<N> hbar;
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            hbar.re[j][j] = pack[P::H_D + j];
            hbar.im[j][j] = 0.0;
SYMPA_UNROLL
            for (int k = j + 1; k < N; ++k) {
                const double hr = pack[P::H_RE + offd_index(N, j, k)], hi = pack[P::H_IM + offd_index(N, j, k)];
                hbar.re[j][k] = hr; hbar.im[j][k] = hi;
                hbar.re[k][j] = hr; hbar.im[k][j] = -hi;
            }
        }
        cmatmul<N>(e, hbar, 2.0, ebar);           // Ebar = 2 E Hbar
    }
    cmatmul_bh<N>(ebar, e, 0.5, a1);              // G = E Hbar E^H
    solve_lh_left<N, CPLX>(l1, ebar);             // Dbar = L1^-H Ebar conj(L2)^-1
    solve_l_right<N, CPLX, true>(l2, ebar);
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
        a2.re[j][j] = pack[P::K_D + j];
        a2.im[j][j] = 0.0;
SYMPA_UNROLL
        for (int k = j + 1; k < N; ++k) {
            const double kr = pack[P::K_RE + offd_index(N, j, k)];
            double ki = 0.0;
            if constexpr (CPLX) ki = -pack[P::K_IM + offd_index(N, j, k)];      // conj(K)
            a2.re[j][k] = kr; a2.im[j][k] = ki;
            a2.re[k][j] = kr; a2.im[k][j] = -ki;
        }
    }
    neg_congruence<N, CPLX>(l1, a1);              // A1bar = -L1^-H G L1^-1
    neg_congruence<N, CPLX>(l2, a2);              // A2bar = -L2^-H conj(K) L2^-1

    if (MODEL == MODEL_UPPER) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) {
                const double dr = 0.5 * (ebar.re[i][j] + ebar.re[j][i]);
                const double di = 0.5 * (ebar.im[i][j] + ebar.im[j][i]);
                g2.re[i][j] = dr;
                g2.im[i][j] = di + a2.re[i][j];
                g1.re[i][j] = -dr;
                g1.im[i][j] = -di + a1.re[i][j];
            }
    } else {
        // Wbar_1 = -Dbar - 2 A1bar W1,  Wbar_2 = Dbar - 2 A2bar W2, then symmetrise each plane
        CMat<N> t1, t2;
        cmatmul<N>(a1, z1, -2.0, t1);
        cmatmul<N>(a2, z2, -2.0, t2);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) {
                t1.re[i][j] -= ebar.re[i][j]; t1.im[i][j] -= ebar.im[i][j];
                t2.re[i][j] += ebar.re[i][j]; t2.im[i][j] += ebar.im[i][j];
            }
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) {
                g1.re[i][j] = 0.5 * (t1.re[i][j] + t1.re[j][i]);
                g1.im[i][j] = 0.5 * (t1.im[i][j] + t1.im[j][i]);
                g2.re[i][j] = 0.5 * (t2.re[i][j] + t2.re[j][i]);
                g2.im[i][j] = 0.5 * (t2.im[i][j] + t2.im[j][i]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2 of the upper model in an order that fits one lane's registers (256 doubles) at n = 8: never more than E / Ebar
// (128) + Hbar (64) + one row (16) at once.  The generic form above keeps E, Ebar, G, K and both factors alive together and
// spills ~530 doubles at n = 8.
//   1. factors, E = L1^-1 (Z2 - Z1) L2^-T; the factors are parked (`park(k, l)`: the caller's LDS, n(n+1)/2 doubles each)
//      -- rebuilding them from Y later was tried first and lost to its loads: a lane-per-row re-read of Y touches 64
//      cache lines per instruction and the rows have left the L2 by then (gradient kernel 468 -> see profiles/r04_n8_backward_split.txt);
//   2. row i = 0 .. n-1:  F = E_i Hbar,  Re G_ij = Re F conj(E_j) for j >= i (rows j >= i of E are still E) -> `put_g`
//      (the caller's workspace, in the slots of Hbar it has read by then),  then Ebar_i = 2 F over E_i;
//   3. Dbar = L1^-T Ebar L2^-1 in place (real factors: the two planes separately), the factors back from the park;
//   4. Re planes: g2 = sym(Re Dbar), g1 = -g2;  Im planes: g2 = sym(Im Dbar) - L2^-T Re K L2^-1,  g1 = -sym(Im Dbar) - L1^-T Re G L1^-1;
//   5. the four planes leave at the very end -> `stage(matrix)` puts a plane into the caller's staging tile (the LDS the factors
//      have left by then), `flush(point, plane, sign)` adds / stores the staged plane: atomics and stores count in the same
//      in-order memory counter as loads, so a load (K, a spilled register) issued behind a plane's 64 atomics waits for all of
//      them -- with the planes interleaved with step 4 the kernel spent half its cycles there (profiles/r04_n8_backward_split.txt).
// pk(k): entry k of the pack (read when it is needed, not before).  Only upper triangles (i <= j) of the emitted matrices
// are meaningful.
// ---------------------------------------------------------------------------------------------
// (sym_congruence_inv_t: siegel_math_bwd.hpp)
template <int N, class Pk, class Park, class Unpark, class PutG, class GetG, class Stage, class Flush>
SYMPA_HD void pair_adjoint_gradient_upper(const CMat<N>& z1, const CMat<N>& z2, Pk&& pk, Park&& park, Unpark&& unpark,
                                          PutG&& put_g, GetG&& get_g, Stage&& stage, Flush&& flush) {
    using P = AdjPack<N, MODEL_UPPER>;
    CMat<N> e;
    double hd[N], hre[N][N], him[N][N];          // Hbar: diagonal, strict upper triangle (j < k)
    {
        Tri<N, false> l1, l2;
        (void)chol_real<N>(z1.im, l1);
        (void)chol_real<N>(z2.im, l2);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) {
                e.re[i][j] = z2.re[i][j] - z1.re[i][j];
                e.im[i][j] = z2.im[i][j] - z1.im[i][j];
            }
        solve_left<N, false>(l1, e);
        park(0, l1);
        // Hbar is asked for HERE, one solve ahead of its first use: one wave per SIMD has nothing else to hide the ~2 us of a
        // workspace read behind (E, L2 and Hbar together still fit: 236 doubles)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            hd[j] = pk(P::H_D + j);
SYMPA_UNROLL
            for (int k = j + 1; k < N; ++k) {
                hre[j][k] = pk(P::H_RE + offd_index(N, j, k));
                him[j][k] = pk(P::H_IM + offd_index(N, j, k));
            }
        }
        solve_right_t<N, false>(l2, e);
        park(1, l2);
    }
    {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) {
            double fr[N], fi[N];
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) {
                double tr = e.re[i][k] * hd[k], ti = e.im[i][k] * hd[k];
SYMPA_UNROLL
                for (int m = 0; m < N; ++m) {
                    if (m == k) continue;
                    // Hbar_mk = (hre[m][k], him[m][k]) for m < k, the conjugate of Hbar_km otherwise
                    const double hr = (m < k) ? hre[m][k] : hre[k][m];
                    const double hi = (m < k) ? him[m][k] : -him[k][m];
                    tr = d_fma(e.re[i][m], hr, tr); tr = d_fma(-e.im[i][m], hi, tr);
                    ti = d_fma(e.re[i][m], hi, ti); ti = d_fma(e.im[i][m], hr, ti);
                }
                fr[k] = tr; fi[k] = ti;
            }
SYMPA_UNROLL
            for (int j = i; j < N; ++j) {
                double g = 0.0;
SYMPA_UNROLL
                for (int k = 0; k < N; ++k) { g = d_fma(fr[k], e.re[j][k], g); g = d_fma(fi[k], e.im[j][k], g); }
                put_g(tri_index(N, i, j), g);
            }
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) { e.re[i][k] = 2.0 * fr[k]; e.im[i][k] = 2.0 * fi[k]; }
        }
    }
    // Dbar = L1^-T Ebar L2^-1
    Tri<N, false> l1, l2;
    unpark(0, l1);
    solve_lh_left<N, false>(l1, e);
    unpark(1, l2);
    solve_l_right<N, false, true>(l2, e);
    double m[N][N], gg[N][N];
    // sym(Dbar) first (Ebar dies) and its real part goes straight into the caller's staging tile (36 doubles fewer through the
    // congruences: with them the tail spilled, and a spilled operand reloaded between two planes' atomics waits for every atomic
    // before it -- the memory counter is in order).  K and G are read when their congruences need them, all before the first
    // plane leaves.
    {
        double di[N][N];
        {
            double dr[N][N];
SYMPA_UNROLL
            for (int i = 0; i < N; ++i)
SYMPA_UNROLL
                for (int j = i; j < N; ++j) {
                    dr[i][j] = 0.5 * (e.re[i][j] + e.re[j][i]);
                    di[i][j] = 0.5 * (e.im[i][j] + e.im[j][i]);
                }
            stage(dr);
        }
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            m[j][j] = pk(P::K_D + j);
SYMPA_UNROLL
            for (int k = j + 1; k < N; ++k) { m[j][k] = pk(P::K_RE + offd_index(N, j, k)); m[k][j] = m[j][k]; }
        }
        sym_congruence_inv_t<N>(l2, m);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = i; j < N; ++j) m[i][j] = di[i][j] - m[i][j];
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = i; j < N; ++j) { gg[i][j] = get_g(tri_index(N, i, j)); gg[j][i] = gg[i][j]; }
        sym_congruence_inv_t<N>(l1, gg);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = i; j < N; ++j) gg[i][j] = -di[i][j] - gg[i][j];
    }
    // the planes leave at the very end, back to back: Re (staged above) to both points, then the two Im planes.  "At the very end"
    // has to be enforced: the first flush reads only the staging tile, so the compiler is free to sink the two congruences above
    // behind its atomics (it did: their spilled operands then waited for 128 atomics each) -- the empty volatile statements below
    // consume the finished planes and, like the atomics, are not reordered with respect to each other.
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = i; j < N; ++j) { SYMPA_PIN(m[i][j]); SYMPA_PIN(gg[i][j]); }
    flush(1, 0, 1.0);
    flush(0, 0, -1.0);
    stage(m);
    flush(1, 1, 1.0);
    stage(gg);
    flush(0, 1, 1.0);
}

}  // namespace sympa
