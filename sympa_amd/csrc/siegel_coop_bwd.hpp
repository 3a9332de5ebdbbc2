// Upper-half and bounded models, backward with sixteen lanes per pair (layout and DPP machinery of spd_coop.hpp / siegel_coop.hpp):
// the training path for 9 <= n <= 16.  The one-lane-per-pair adjoint keeps E, H, V and four more n x n complex matrices
// of a pair in scratch memory (siegel_bwd_rolled.hip: 5 M pairs/s at n = 10, 1 M at n = 16); here lane r of a group of
// sixteen owns row r (or column r, where noted) of every matrix of its pair and everything stays in registers.
//
// Formulas: siegel_math_bwd.hpp's header, in the form that never touches the adjoint of the Cholesky factorisation:
//   E = L1^-1 D L2^-T (held by columns, as in the forward kernel),  H = E^H E = Q T' Q^H,  T' = Phi T Phi^H,  T = Z Lambda Z^T
//   V = Q Phi Z                      complex Householder with the reflectors KEPT (lane r: component r of every v_k),
//                                    the phases Phi that make the tridiagonal real, tql2 with the rotations applied to
//                                    the lane's own row of Z (spd_coop_bwd.hpp), back-transformation in column layout
//   U = E V                          column c of U from column c of V and the rows of E^T, both already in place;
//                                    lambda_c = ||U[:, c]||^2 (Rayleigh quotient: relative accuracy for small lambda)
//   Ebar = 2 U diag(phi) V^H,  Dbar = L1^-T Ebar L2^-1
//   A1bar = -L1^-T Re(U diag(phi) U^H) L1^-1,  A2bar = -L2^-T Re(V diag(phi lambda) V^H) L2^-1
//   X2bar = sym Re Dbar,  Y2bar = sym Im Dbar + A2bar,  X1bar = -sym Re Dbar,  Y1bar = -sym Im Dbar + A1bar
// Bounded model: E = C1^-1 D C2^-T with the complex factors of I - W_k W_k^H = C_k C_k^H (siegel_coop.hpp), the same
// eigen part, then  Dbar = C1^-H Ebar conj(C2)^-1,  A1bar = -C1^-H (U diag(phi) U^H) C1^-1,
// A2bar = -C2^-H conj(V diag(phi lambda) V^H) C2^-1,  W1bar = -Dbar - 2 A1bar W1,  W2bar = Dbar - 2 A2bar W2, each plane
// symmetrised.
// Checked on the GPU against the one-lane-per-pair kernel (SYMPA_FLAG_GENERIC), the g++ build of the same adjoint and the
// reference-autograd goldens.
#pragma once

#include "siegel_coop.hpp"
#include "siegel_math_bwd.hpp"
#include "spd_coop_bwd.hpp"

namespace siegel_coop {

// Complex Householder tridiagonalisation of the Hermitian H held one row per lane, reflectors kept:
//   d[k] = T'[k][k], (br, bi)[k] = T'[k+1][k] group-uniform;  (vr, vi)[k] = my component of reflector k;  bk[k] = beta_k
//   (P_k = I - beta_k v_k v_k^H,  T' = P_{M-3} ... P_0 H P_0 ... P_{M-3})
template <int M>
__device__ __forceinline__ void ctridiagonalize_keep(double (&hr)[M], double (&hi)[M], const int r, double (&d)[M],
                                                     double (&br)[M], double (&bi)[M], double (&vr)[M], double (&vi)[M],
                                                     double (&bk)[M]) {
    sfor<0, M - 2>([&](auto K) {
        constexpr int k = K;
        const double cr = settle(hr[k]), ci = settle(hi[k]);           // my element of column k: H[me][k]
        const double x0r = bcast<k + 1>(cr), x0i = bcast<k + 1>(ci);
        const double dk = bcast<k>(cr);
        const double t2 = (r > k + 1 && r < M) ? sympa::d_fma(cr, cr, ci * ci) : 0.0;
        const double s2 = group_sum(t2);
        const double x02 = sympa::d_fma(x0r, x0r, x0i * x0i);
        const double n2 = x02 + s2;
        const double nx = sympa::d_sqrt(n2);
        const double ix0 = sympa::d_rsqrt(x02 + sympa::TINY);
        const double ax0 = x02 * ix0;
        const bool x0zero = !(x02 > 0.0);
        const double pr = x0zero ? 1.0 : x0r * ix0, pi = x0zero ? 0.0 : x0i * ix0;      // phase of x0
        const double v0r = pr * (ax0 + nx), v0i = pi * (ax0 + nx);                      // v = x + phase ||x|| e1
        const bool reflect = s2 > 0.0;
        const double beta = reflect ? sympa::d_rcp(nx * (nx + ax0)) : 0.0;              // 2 / ||v||^2
        d[k] = dk;
        br[k] = reflect ? -pr * nx : x0r;                                               // P x = -phase ||x|| e1
        bi[k] = reflect ? -pi * nx : x0i;
        const double wr = settle((r <= k || r >= M) ? 0.0 : ((r == k + 1) ? v0r : cr));
        const double wi = settle((r <= k || r >= M) ? 0.0 : ((r == k + 1) ? v0i : ci));
        vr[k] = wr; vi[k] = wi; bk[k] = beta;
        // p = beta H v:  p_i = sum_j H[i][j] v_j
        double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
        sfor<k + 1, M>([&](auto J) {
            constexpr int j = J;
            fmac_bc<j>(p0, wr, hr[j]);
            fnmac_bc<j>(p1, wi, hi[j]);
            fmac_bc<j>(q0, wi, hr[j]);
            fmac_bc<j>(q1, wr, hi[j]);
        });
        const double pr_ = (r <= k || r >= M) ? 0.0 : beta * (p0 + p1);
        const double pi_ = (r <= k || r >= M) ? 0.0 : beta * (q0 + q1);
        const double kk = 0.5 * beta * group_sum(sympa::d_fma(wr, pr_, wi * pi_));      // Re(v^H p) beta / 2
        const double qr = settle(sympa::d_fma(-kk, wr, pr_));
        const double qi = settle(sympa::d_fma(-kk, wi, pi_));
        // H <- H - v q^H - q v^H:   H[i][j] -= v_i conj(q_j) + q_i conj(v_j)
        sfor<k + 1, M>([&](auto J) {
            constexpr int j = J;
            fnmac_bc<j>(hr[j], qr, wr);
            fnmac_bc<j>(hr[j], qi, wi);
            fnmac_bc<j>(hr[j], wr, qr);
            fnmac_bc<j>(hr[j], wi, qi);
            fnmac_bc<j>(hi[j], qr, wi);
            fmac_bc<j>(hi[j], qi, wr);
            fnmac_bc<j>(hi[j], wr, qi);
            fmac_bc<j>(hi[j], wi, qr);
        });
    });
    if constexpr (M >= 2) {
        const double sr = settle(hr[M - 2]), si = settle(hi[M - 2]);
        d[M - 2] = bcast<M - 2>(sr);
        br[M - 2] = bcast<M - 1>(sr);                // T'[M-1][M-2] = H[M-1][M-2]: register M-2 of lane M-1
        bi[M - 2] = bcast<M - 1>(si);
        vr[M - 2] = 0.0; vi[M - 2] = 0.0; bk[M - 2] = 0.0;
    }
    d[M - 1] = bcast<M - 1>(settle(hr[M - 1]));
    br[M - 1] = 0.0; bi[M - 1] = 0.0; vr[M - 1] = 0.0; vi[M - 1] = 0.0; bk[M - 1] = 0.0;
}

// (zr, zi) = my COLUMN of Phi Z;  z <- P_0 ... P_{M-3} z,  P_k = I - beta_k v_k v_k^H, component r of v_k in lane r
template <int M>
__device__ __forceinline__ void cback_transform_columns(double (&zr)[M], double (&zi)[M], double (&vr)[M], double (&vi)[M],
                                                        const double (&bk)[M]) {
    sfor<0, M - 2>([&](auto KK) {
        constexpr int k = M - 3 - KK;
        const double sr = settle(vr[k]), si = settle(vi[k]);
        // tau = beta v^H z
        double t0 = 0.0, t1 = 0.0, u0 = 0.0, u1 = 0.0;
        sfor<k + 1, M>([&](auto R) {
            constexpr int rr = R;
            fmac_bc<rr>(t0, sr, zr[rr]);
            fmac_bc<rr>(t1, si, zi[rr]);
            fmac_bc<rr>(u0, sr, zi[rr]);
            fnmac_bc<rr>(u1, si, zr[rr]);
        });
        const double tr = bk[k] * (t0 + t1), ti = bk[k] * (u0 + u1);
        // z -= v tau
        sfor<k + 1, M>([&](auto R) {
            constexpr int rr = R;
            fnmac_bc<rr>(zr[rr], sr, tr);
            fmac_bc<rr>(zr[rr], si, ti);
            fnmac_bc<rr>(zi[rr], sr, ti);
            fnmac_bc<rr>(zi[rr], si, tr);
        });
    });
}

// U^T rows (= my column of U = E V) from my column of V and the rows of E^T:  U[j][me] = sum_k E^T[k][j] V[k][me]
template <int M>
__device__ __forceinline__ void ut_from_columns(double (&etr)[M], double (&eti)[M], const double (&vcr)[M],
                                                const double (&vci)[M], double (&utr)[M], double (&uti)[M]) {
    sfor<0, M>([&](auto J) { etr[J] = settle(etr[J]); eti[J] = settle(eti[J]); });
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        sfor<0, M>([&](auto K) {
            constexpr int k = K;
            fmac_bc<k>(a0, etr[j], vcr[k]);
            fnmac_bc<k>(a1, eti[j], vci[k]);
            fmac_bc<k>(b0, etr[j], vci[k]);
            fmac_bc<k>(b1, eti[j], vcr[k]);
        });
        utr[j] = a0 + a1;
        uti[j] = b0 + b1;
    });
}

// rows of  A diag(s) B^H  from the rows of A (mine) and of B (broadcast):  out[me][j] = sum_c A[me][c] s_c conj(B[j][c])
template <int M, bool REAL_ONLY>
__device__ __forceinline__ void adbh_rows(const double (&ar)[M], const double (&ai)[M], const double (&s)[M], double (&brw)[M],
                                          double (&biw)[M], double (&outr)[M], double (&outi)[M]) {
    double sr[M], si[M];
    sfor<0, M>([&](auto C) {
        sr[C] = ar[C] * s[C]; si[C] = ai[C] * s[C];
        brw[C] = settle(brw[C]); biw[C] = settle(biw[C]);
    });
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        sfor<0, M>([&](auto C) {
            constexpr int c = C;
            fmac_bc<j>(a0, brw[c], sr[c]);
            fmac_bc<j>(a1, biw[c], si[c]);
            if constexpr (!REAL_ONLY) {
                fmac_bc<j>(b0, brw[c], si[c]);
                fnmac_bc<j>(b1, biw[c], sr[c]);
            }
        });
        outr[j] = a0 + a1;
        if constexpr (!REAL_ONLY) outi[j] = b0 + b1;
    });
}

// a <- a M^-1 for complex rows held one per lane, M = C (CONJ = false) or conj(C) (CONJ = true), C lower triangular with
// a real diagonal (rd = 1 / diag):  a'[j] = (a[j] - sum_{k>j} a'[k] M[k][j]) / C[j][j];  C[k][j] is register j of lane k.
template <int M, bool CONJ>
__device__ __forceinline__ void csolve_right_l(double (&ar)[M], double (&ai)[M], const double (&cr)[M], const double (&ci)[M],
                                               const double (&rd)[M]) {
    sfor<0, M>([&](auto JJ) {
        constexpr int j = M - 1 - JJ;
        sfor<j + 1, M>([&](auto K) {
            constexpr int k = K;
            // a'[k] (mr + i mi),  mi = -ci for conj(C)
            fnmac_bc<k>(ar[j], cr[j], ar[k]);
            fnmac_bc<k>(ai[j], cr[j], ai[k]);
            if constexpr (CONJ) {
                fnmac_bc<k>(ar[j], ci[j], ai[k]);
                fmac_bc<k>(ai[j], ci[j], ar[k]);
            } else {
                fmac_bc<k>(ar[j], ci[j], ai[k]);
                fnmac_bc<k>(ai[j], ci[j], ar[k]);
            }
        });
        ar[j] *= rd[j];
        ai[j] *= rd[j];
    });
}

// rows of T = A W from the rows of A (mine) and of W (broadcast):  T[me][j] = sum_k A[me][k] W[k][j]
template <int M>
__device__ __forceinline__ void cmatmul_rows(const double (&ar)[M], const double (&ai)[M], double (&wr)[M], double (&wi)[M],
                                             double (&tr)[M], double (&ti)[M]) {
    sfor<0, M>([&](auto J) { wr[J] = settle(wr[J]); wi[J] = settle(wi[J]); });
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        double t0 = 0.0, t1 = 0.0, u0 = 0.0, u1 = 0.0;
        sfor<0, M>([&](auto K) {
            constexpr int k = K;
            fmac_bc<k>(t0, wr[j], ar[k]);
            fnmac_bc<k>(t1, wi[j], ai[k]);
            fmac_bc<k>(u0, wi[j], ar[k]);
            fmac_bc<k>(u1, wr[j], ai[k]);
        });
        tr[j] = t0 + t1;
        ti[j] = u0 + u1;
    });
}

// Hermitian p (rows) <- C^-H p C^-1 (rows), complex factor:  q = p C^-1, transpose, q^T conj(C)^-1 = (C^-H q)^T = conj of the
// rows of the Hermitian result
template <int M>
__device__ __forceinline__ void ccongruence_inv_h_rows(double (&pr)[M], double (&pi)[M], double (&cr)[M], double (&ci)[M],
                                                       const double (&rd)[M], double* __restrict__ tbuf, const int r) {
    sfor<0, M>([&](auto J) { cr[J] = settle(cr[J]); ci[J] = settle(ci[J]); });
    csolve_right_l<M, false>(pr, pi, cr, ci, rd);
    double tr[M], ti[M];
    spd_coop::transpose_rows(pr, tr, tbuf, r);
    spd_coop::transpose_rows(pi, ti, tbuf, r);
    csolve_right_l<M, true>(tr, ti, cr, ci, rd);
    sfor<0, M>([&](auto J) { pr[J] = tr[J]; pi[J] = -ti[J]; });
}

}  // namespace siegel_coop
