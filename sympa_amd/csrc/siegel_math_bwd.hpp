// Analytic backward of the Siegel distance, per pair, same compile-twice scheme as
// siegel_math.hpp (hipcc for the kernels, g++ for tests/hostsim).
//
// It replaces what the reference obtains from torch autograd through eigh / inverse / bmm
// (runner.py:105 `loss.backward()` over siegel_manifold.py:41-72): given go = dLoss/d(dist) it returns
// the SYMMETRIC matrix gradients dLoss/dZ1, dLoss/dZ2 (real and imaginary planes), i.e. exactly the
// tensors autograd puts into `embeds.grad` rows (the reference's gradients are symmetric to 1e-12,
// tests/golden/autograd_*.npz), plus the gradient of the wsum weights.
//
// With E = L1^-1 D L2^-T (upper; D = Z2 - Z1, Y_k = L_k L_k^T) or E = C1^-1 D C2^-T (bounded;
// D = W2 - W1, A_k = I - W_k W_k^H = C_k C_k^H), H = E^H E = V diag(lambda) V^H, out = m(v(lambda)):
//   phi_i  = go * dm/dv_i * dv/dlambda_i                      dv/dlambda' = 1/sqrt(lambda'(1+lambda'))
//   Hbar   = V diag(phi) V^H,   K = V diag(phi lambda) V^H = Hbar H
//   Ebar   = 2 E Hbar,          G = E Hbar E^H
//   Dbar   = L1^-H Ebar conj(L2)^-1                            (conj only matters for complex factors)
//   A1bar  = -L1^-H G L1^-1,   A2bar = -L2^-H conj(K) L2^-1
//     (the lambda_i are the eigenvalues of the pencils (D conj(A2)^-1 D^H, A1) and (D^H A1^-1 D, conj(A2)), so
//      d lambda_i = -lambda_i u_i^H dA u_i with u_i = L^-H (singular vector): the Cholesky factors only enter through
//      these congruences and no adjoint of the factorisation itself is needed)
//   upper  : Ybar_k += Abar_k (real);    X2bar += Re Dbar, Y2bar += Im Dbar, X1bar -= ..., Y1bar -= ...
//   bounded: Wbar_k  = -/+ Dbar - 2 Abar_k W_k
// and finally every plane is symmetrised.  Derivation and the numpy prototype that was checked against
// the reference's autograd to 1e-11: DESIGN.md section 8.  Eigenvectors come from the same Jacobi
// iteration with the rotations accumulated, run to ||off|| <= 1e-11 ||diag|| (no finishing shortcut).
#pragma once

#include "siegel_math.hpp"

namespace sympa {

// ---------------------------------------------------------------------------------------------
// Jacobi with eigenvectors: H <- J^H H J, V <- V J.
// ---------------------------------------------------------------------------------------------
template <int N>
SYMPA_HD void jacobi_rotate_vec(Herm<N>& h, CMat<N>& v, const int p, const int q) {
    const double br = h.re[p][q], bi = h.im[p][q];
    const double a2 = d_fma(br, br, bi * bi);
    const double delta = h.d[q] - h.d[p];
    const double ad = fabs(delta) + 1e-150;
    const double qr = d_rsqrt(d_fma(ad, ad, 4.0 * a2));
    const double c2 = d_fma(0.5 * ad, qr, 0.5);
    const double ic = d_rsqrt(c2);
    const double c = c2 * ic;
    const double cu = copysign(qr, delta) * ic;
    const double ua2 = (cu * ic) * a2;
    const double wr = cu * br, wi = cu * bi;
    h.d[p] -= ua2;
    h.d[q] += ua2;
    h.re[p][q] = 0.0;
    h.im[p][q] = 0.0;
SYMPA_UNROLL
    for (int k = 0; k < N; ++k) {
        if (k == p || k == q) continue;
        double xr, xi, yr, yi;
        if (k < p) { xr = h.re[k][p]; xi = h.im[k][p]; } else { xr = h.re[p][k]; xi = -h.im[p][k]; }
        if (k < q) { yr = h.re[k][q]; yi = h.im[k][q]; } else { yr = h.re[q][k]; yi = -h.im[q][k]; }
        const double nxr = d_fma(-wi, yi, d_fma(-wr, yr, c * xr));
        const double nxi = d_fma(wi, yr, d_fma(-wr, yi, c * xi));
        const double nyr = d_fma(-wi, xi, d_fma(wr, xr, c * yr));
        const double nyi = d_fma(wi, xr, d_fma(wr, xi, c * yi));
        if (k < p) { h.re[k][p] = nxr; h.im[k][p] = nxi; } else { h.re[p][k] = nxr; h.im[p][k] = -nxi; }
        if (k < q) { h.re[k][q] = nyr; h.im[k][q] = nyi; } else { h.re[q][k] = nyr; h.im[q][k] = -nyi; }
    }
SYMPA_UNROLL
    for (int k = 0; k < N; ++k) {   // columns p, q of V
        const double xr = v.re[k][p], xi = v.im[k][p], yr = v.re[k][q], yi = v.im[k][q];
        v.re[k][p] = d_fma(-wi, yi, d_fma(-wr, yr, c * xr));
        v.im[k][p] = d_fma(wi, yr, d_fma(-wr, yi, c * xi));
        v.re[k][q] = d_fma(-wi, xi, d_fma(wr, xr, c * yr));
        v.im[k][q] = d_fma(wi, xr, d_fma(wr, xi, c * yi));
    }
}

constexpr double JACOBI_VEC_TOL2 = 1e-22;

template <int N>
SYMPA_HD bool herm_eigen_vectors(Herm<N>& h, CMat<N>& v) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) { v.re[i][j] = (i == j) ? 1.0 : 0.0; v.im[i][j] = 0.0; }
    if (N == 1) return true;
    bool conv = false;
    for (int sweep = 0; sweep < JACOBI_MAX_SWEEPS; ++sweep) {
        double off2, diag2;
        herm_norms<N>(h, off2, diag2);
        conv = !(off2 > JACOBI_VEC_TOL2 * diag2);
        if (wave_all(conv)) break;
        if (N == 4) {
            jacobi_rotate_vec<N>(h, v, 0, 1); jacobi_rotate_vec<N>(h, v, 2, 3);
            jacobi_rotate_vec<N>(h, v, 0, 2); jacobi_rotate_vec<N>(h, v, 1, 3);
            jacobi_rotate_vec<N>(h, v, 0, 3); jacobi_rotate_vec<N>(h, v, 1, 2);
        } else {
SYMPA_UNROLL
            for (int p = 0; p < N - 1; ++p) {
SYMPA_UNROLL
                for (int q = p + 1; q < N; ++q) jacobi_rotate_vec<N>(h, v, p, q);
            }
        }
    }
    return conv;
}

// ---------------------------------------------------------------------------------------------
// n >= 5: eigenvectors through Householder tridiagonalisation (reflectors kept) + implicit QL with the
// rotations accumulated (EISPACK tql2 / LAPACK dsteqr) + back-transformation.  A cyclic Jacobi iteration with
// vectors costs ~200 instructions per rotation, 28 rotations per sweep, 7-8 sweeps at n = 8 (~40 k
// instructions, half of the whole adjoint); this route costs ~9 k.  Per-lane active block [L, m]: the QL sweep
// of stage L runs over the static index range [L, N-2] and a lane's steps outside its block are identities
// (c = 1, s = 0) selected by predicates -- with 4 N instructions of eigenvector update per step the bookkeeping
// (~40 instructions) is a small part, unlike in the eigenvalue-only iteration (siegel_math.hpp, lockstep QL).
// The eigenvalues that come out are accurate to eps ||H||; the caller refines them to RELATIVE accuracy with the
// Rayleigh quotients lambda_i = ||E v_i||^2 (H = E^H E), which the adjoint needs for small lambda (d v / d lambda
// ~ lambda^-1/2).
// ---------------------------------------------------------------------------------------------
template <int N>
SYMPA_HD void herm_tridiagonalize_keep(Herm<N>& h, double (&a)[N], double (&e)[N], double (&phr)[N], double (&phi)[N],
                                       CMat<N>& refl, double (&beta)[N]) {
SYMPA_UNROLL
    for (int k = 0; k < N - 2; ++k) {
        double vr[N], vi[N];
        double sig2 = 0.0;
SYMPA_UNROLL
        for (int i = k + 1; i < N; ++i) {
            herm_get<N>(h, i, k, vr[i], vi[i]);
            if (i > k + 1) sig2 = d_fma(vr[i], vr[i], d_fma(vi[i], vi[i], sig2));
        }
        const double x02 = d_fma(vr[k + 1], vr[k + 1], vi[k + 1] * vi[k + 1]);
        const double n2 = x02 + sig2;
        a[k] = h.d[k];
        const double nx = d_sqrt(n2);
        const double ix0 = d_rsqrt(x02 + TINY);
        const double ax0 = x02 * ix0;
        const bool x0zero = !(x02 > 0.0);
        const double pr = x0zero ? 1.0 : vr[k + 1] * ix0, pi = x0zero ? 0.0 : vi[k + 1] * ix0;   // phase of x0
        const bool reflect = sig2 > 0.0;
        // T[k+1][k] = -phase ||x|| after the reflection, x0 itself when there is nothing to eliminate
        e[k] = reflect ? nx : ax0;
        phr[k] = reflect ? -pr : pr;
        phi[k] = reflect ? -pi : pi;
        vr[k + 1] = pr * (ax0 + nx);
        vi[k + 1] = pi * (ax0 + nx);
        const double bt = reflect ? d_rcp(nx * (nx + ax0)) : 0.0;
        beta[k] = bt;
SYMPA_UNROLL
        for (int i = k + 1; i < N; ++i) { refl.re[i][k] = vr[i]; refl.im[i][k] = vi[i]; }
        double qr[N], qi[N];
        double kk = 0.0;
SYMPA_UNROLL
        for (int i = k + 1; i < N; ++i) {
            double tr = h.d[i] * vr[i], ti = h.d[i] * vi[i];
SYMPA_UNROLL
            for (int j = k + 1; j < N; ++j) {
                if (j == i) continue;
                double ar, ai;
                herm_get<N>(h, i, j, ar, ai);
                tr = d_fma(ar, vr[j], tr); tr = d_fma(-ai, vi[j], tr);
                ti = d_fma(ar, vi[j], ti); ti = d_fma(ai, vr[j], ti);
            }
            qr[i] = bt * tr; qi[i] = bt * ti;
            kk = d_fma(vr[i], qr[i], kk); kk = d_fma(vi[i], qi[i], kk);
        }
        kk *= 0.5 * bt;
SYMPA_UNROLL
        for (int i = k + 1; i < N; ++i) { qr[i] = d_fma(-kk, vr[i], qr[i]); qi[i] = d_fma(-kk, vi[i], qi[i]); }
SYMPA_UNROLL
        for (int i = k + 1; i < N; ++i) {
            h.d[i] -= 2.0 * d_fma(vr[i], qr[i], vi[i] * qi[i]);
SYMPA_UNROLL
            for (int j = i + 1; j < N; ++j) {
                double tr = d_fma(vr[i], qr[j], vi[i] * qi[j]);
                double ti = d_fma(vi[i], qr[j], -vr[i] * qi[j]);
                tr = d_fma(qr[i], vr[j], d_fma(qi[i], vi[j], tr));
                ti = d_fma(qi[i], vr[j], d_fma(-qr[i], vi[j], ti));
                h.re[i][j] -= tr;
                h.im[i][j] -= ti;
            }
        }
    }
    a[N - 2] = h.d[N - 2];
    a[N - 1] = h.d[N - 1];
    // T[N-1][N-2] = conj of the stored upper element (N-2, N-1)
    const double lr = h.re[N - 2][N - 1], li = -h.im[N - 2][N - 1];
    const double l2 = d_fma(lr, lr, li * li);
    const double il = d_rsqrt(l2 + TINY);
    const bool lz = !(l2 > 0.0);
    e[N - 2] = l2 * il;
    phr[N - 2] = lz ? 1.0 : lr * il;
    phi[N - 2] = lz ? 0.0 : li * il;
    e[N - 1] = 0.0;
}

// Implicit QL with accumulated rotations on the real symmetric tridiagonal (d, e): e[i] couples i and i + 1.
// z must hold the identity on entry; on exit z's columns are the eigenvectors and d the eigenvalues.
template <int N>
SYMPA_HD bool tridiag_ql_vectors(double (&d)[N], double (&e)[N], double (&z)[N][N]) {
    bool all_ok = true;
SYMPA_UNROLL
    for (int L = 0; L < N - 1; ++L) {
        bool conv = false;
        for (int it = 0; it < 50; ++it) {
            // m = the first index >= L whose off-diagonal is negligible (e[N-1] = 0): the block is [L, m]
            int m = N - 1;
SYMPA_UNROLL
            for (int i = N - 2; i >= L; --i) {
                const bool negl = ql_negligible(e[i] * e[i], d[i], d[i + 1]);
                e[i] = negl ? 0.0 : e[i];
                m = negl ? i : m;
            }
            conv = (m == L);
            if (wave_all(conv)) break;
            // Wilkinson shift from the leading 2 x 2 of the block (tql2): g = d[m] - d[L] + e[L] / (g0 + sign(r0, g0))
            const double el = conv ? 1.0 : e[L];
            const double g0 = 0.5 * (d[L + 1] - d[L]) * d_rcp(el);
            const double r0 = d_sqrt(d_fma(g0, g0, 1.0));
            const double shift = el * d_rcp(g0 + copysign(r0, g0)) - d[L];
            double c = 1.0, s = 1.0, p = 0.0, g = 0.0;
SYMPA_UNROLL
            for (int i = N - 2; i >= L; --i) {
                const bool active = !conv && (i < m);
                const bool start = (m == i + 1);
                g = start ? d[i + 1] + shift : g;
                s = start ? 1.0 : s;
                c = start ? 1.0 : c;
                p = start ? 0.0 : p;
                const double f = s * e[i];
                const double b = c * e[i];
                const double r2 = d_fma(f, f, g * g);
                const double ir = d_rsqrt(r2 + TINY);
                const double r = r2 * ir;
                const bool rzero = !(r2 > 0.0);
                const double sn = rzero ? 0.0 : f * ir;
                const double cn = rzero ? 1.0 : g * ir;
                if (i + 1 <= N - 2) e[i + 1] = (active && !start) ? r : e[i + 1];
                const double g2 = d[i + 1] - p;
                const double rr = d_fma(d[i] - g2, sn, 2.0 * cn * b);
                const double pn = sn * rr;
                d[i + 1] = active ? g2 + pn : d[i + 1];
                const double gn = d_fma(cn, rr, -b);
                const double ce = active ? cn : 1.0, se = active ? sn : 0.0;
SYMPA_UNROLL
                for (int k = 0; k < N; ++k) {
                    const double zf = z[k][i + 1];
                    z[k][i + 1] = d_fma(se, z[k][i], ce * zf);
                    z[k][i] = d_fma(ce, z[k][i], -se * zf);
                }
                s = active ? sn : s;
                c = active ? cn : c;
                p = active ? pn : p;
                g = active ? gn : g;
            }
            d[L] = conv ? d[L] : d[L] - p;
            e[L] = conv ? e[L] : g;
        }
        all_ok = all_ok && conv;
    }
    return all_ok;
}

// Eigen-decomposition of the Hermitian h: eigenvalues into h.d, eigenvectors into the columns of v.
template <int N>
SYMPA_HD bool herm_eigen_vectors_ql(Herm<N>& h, CMat<N>& v) {
    double a[N], e[N], phr[N], phi[N], beta[N];
    CMat<N> refl;
    herm_tridiagonalize_keep<N>(h, a, e, phr, phi, refl, beta);
    double z[N][N];
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) z[i][j] = (i == j) ? 1.0 : 0.0;
    const bool ok = tridiag_ql_vectors<N>(a, e, z);
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
    for (int i = 0; i < N; ++i) h.d[i] = a[i];
    return ok;
}

// out_jk = sum_i s_i V_ji conj(V_ki)    (full Hermitian matrix, both triangles)
template <int N>
SYMPA_HD void herm_from_eig(const CMat<N>& v, const double (&s)[N], CMat<N>& out) {
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
SYMPA_UNROLL
        for (int k = j; k < N; ++k) {
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int i = 0; i < N; ++i) {
                const double ar = s[i] * v.re[j][i], ai = s[i] * v.im[j][i];
                tr = d_fma(ar, v.re[k][i], tr);
                tr = d_fma(ai, v.im[k][i], tr);
                ti = d_fma(ai, v.re[k][i], ti);
                ti = d_fma(-ar, v.im[k][i], ti);
            }
            out.re[j][k] = tr; out.im[j][k] = (j == k) ? 0.0 : ti;
            out.re[k][j] = tr; out.im[k][j] = (j == k) ? 0.0 : -ti;
        }
    }
}

// c = alpha * a * b
template <int N>
SYMPA_HD void cmatmul(const CMat<N>& a, const CMat<N>& b, double alpha, CMat<N>& c) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) {
                tr = d_fma(a.re[i][k], b.re[k][j], tr);
                tr = d_fma(-a.im[i][k], b.im[k][j], tr);
                ti = d_fma(a.re[i][k], b.im[k][j], ti);
                ti = d_fma(a.im[i][k], b.re[k][j], ti);
            }
            c.re[i][j] = alpha * tr;
            c.im[i][j] = alpha * ti;
        }
}

// c = alpha * a * b^H
template <int N>
SYMPA_HD void cmatmul_bh(const CMat<N>& a, const CMat<N>& b, double alpha, CMat<N>& c) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) {   // a_ik conj(b_jk)
                tr = d_fma(a.re[i][k], b.re[j][k], tr);
                tr = d_fma(a.im[i][k], b.im[j][k], tr);
                ti = d_fma(a.im[i][k], b.re[j][k], ti);
                ti = d_fma(-a.re[i][k], b.im[j][k], ti);
            }
            c.re[i][j] = alpha * tr;
            c.im[i][j] = alpha * ti;
        }
}

// Element (i, j) of the lower-triangular factor as a complex number (diagonal = 1 / rdiag, real).
template <int N, bool COMPLEX>
SYMPA_HD void tri_elem(const Tri<N, COMPLEX>& l, const double (&diag)[N], int i, int j, double& re, double& im) {
    if (i == j) { re = diag[i]; im = 0.0; }
    else { re = l.re[i][j]; im = COMPLEX ? l.im[i][j] : 0.0; }
}

// X <- L^-H X   (back substitution, column by column)
template <int N, bool COMPLEX>
SYMPA_HD void solve_lh_left(const Tri<N, COMPLEX>& l, CMat<N>& x) {
SYMPA_UNROLL
    for (int c = 0; c < N; ++c) {
SYMPA_UNROLL
        for (int i = N - 1; i >= 0; --i) {
            double tr = x.re[i][c], ti = x.im[i][c];
SYMPA_UNROLL
            for (int k = i + 1; k < N; ++k) {   // (L^H)_ik = conj(L_ki)
                tr = d_fma(-l.re[k][i], x.re[k][c], tr);
                ti = d_fma(-l.re[k][i], x.im[k][c], ti);
                if (COMPLEX) {
                    tr = d_fma(-l.im[k][i], x.im[k][c], tr);
                    ti = d_fma(l.im[k][i], x.re[k][c], ti);
                }
            }
            x.re[i][c] = tr * l.rdiag[i];
            x.im[i][c] = ti * l.rdiag[i];
        }
    }
}

// X <- X M^-1 with M = L (CONJ = false) or conj(L) (CONJ = true); row by row, columns from the right
template <int N, bool COMPLEX, bool CONJ>
SYMPA_HD void solve_l_right(const Tri<N, COMPLEX>& l, CMat<N>& x) {
SYMPA_UNROLL
    for (int r = 0; r < N; ++r) {
SYMPA_UNROLL
        for (int j = N - 1; j >= 0; --j) {
            double tr = x.re[r][j], ti = x.im[r][j];
SYMPA_UNROLL
            for (int k = j + 1; k < N; ++k) {   // minus x_rk M_kj
                const double mi = COMPLEX ? (CONJ ? -l.im[k][j] : l.im[k][j]) : 0.0;
                tr = d_fma(-x.re[r][k], l.re[k][j], tr);
                ti = d_fma(-x.im[r][k], l.re[k][j], ti);
                if (COMPLEX) {
                    tr = d_fma(x.im[r][k], mi, tr);
                    ti = d_fma(-x.re[r][k], mi, ti);
                }
            }
            x.re[r][j] = tr * l.rdiag[j];
            x.im[r][j] = ti * l.rdiag[j];
        }
    }
}

// m <- -herm(L^-H m L^-1) for a Hermitian m (real factor: the two planes stay independent and a caller that only reads
// the real plane leaves the imaginary one to dead-code elimination)
template <int N, bool COMPLEX>
SYMPA_HD void neg_congruence(const Tri<N, COMPLEX>& l, CMat<N>& m) {
    solve_lh_left<N, COMPLEX>(l, m);
    solve_l_right<N, COMPLEX, false>(l, m);
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = i; j < N; ++j) {
            const double re = -0.5 * (m.re[i][j] + m.re[j][i]);
            const double im = COMPLEX ? -0.5 * (m.im[i][j] - m.im[j][i]) : 0.0;
            m.re[i][j] = re; m.re[j][i] = re;
            m.im[i][j] = im; m.im[j][i] = -im;
        }
}

// Cholesky adjoint (the route through the adjoint of the factorisation itself; equal to neg_congruence on L^-H M with
// scale = -2 -- kept for n = 8, where this instruction order happens to spill less, see pair_backward).
// In: L (factor), M (full matrix; only tril(L^-H M) enters), scale.  Computes
//   Lbar = scale * tril(L^-H M),  P = Phi(L^H Lbar),  Abar = herm(L^-H P L^-1)
// and returns Abar as a full Hermitian matrix.
template <int N, bool COMPLEX>
SYMPA_HD void chol_adjoint(const Tri<N, COMPLEX>& l, CMat<N>& m, double scale, CMat<N>& abar) {
    double diag[N];
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) diag[i] = d_rcp(l.rdiag[i]);
    solve_lh_left<N, COMPLEX>(l, m);           // m = L^-H M
    // Lbar = scale * tril(m)
    // P = Phi(L^H Lbar):  P_ij = sum_{k >= max(i,j)} conj(L_ki) Lbar_kj,  for i >= j
    CMat<N> p;
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            double tr = 0.0, ti = 0.0;
            if (i >= j) {
SYMPA_UNROLL
                for (int k = i; k < N; ++k) {
                    double lr, li;
                    tri_elem<N, COMPLEX>(l, diag, k, i, lr, li);
                    const double br = scale * m.re[k][j], bi = COMPLEX ? scale * m.im[k][j] : 0.0;
                    // conj(L_ki) * Lbar_kj
                    tr = d_fma(lr, br, tr);
                    if (COMPLEX) {
                        tr = d_fma(li, bi, tr);
                        ti = d_fma(lr, bi, ti);
                        ti = d_fma(-li, br, ti);
                    }
                }
                if (i == j) { tr *= 0.5; ti = 0.0; }
            }
            p.re[i][j] = tr;
            p.im[i][j] = ti;
        }
    // S = L^-H P L^-1
    solve_lh_left<N, COMPLEX>(l, p);
    solve_l_right<N, COMPLEX, false>(l, p);
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            abar.re[i][j] = 0.5 * (p.re[i][j] + p.re[j][i]);
            abar.im[i][j] = COMPLEX ? 0.5 * (p.im[i][j] - p.im[j][i]) : 0.0;
        }
}

// ---------------------------------------------------------------------------------------------
// The scalar part of the adjoint: eigenvalues lambda of H -> value of the metric and the spectral weights
//   phi_i = go * d out / d lambda_i,   philam_i = phi_i lambda_i,   gw[k] += go * d out / d w_k  (wsum)
// (shared by the one-pair-per-lane adjoint below and the sixteen-lanes-per-pair one, siegel_coop_bwd.hpp)
// ---------------------------------------------------------------------------------------------
template <int N, int MODEL>
SYMPA_HD double spectral_adjoint(const double (&lam_in)[N], int metric, const double* __restrict__ w, double inv_eps,
                                 double go, double (&phi)[N], double (&philam)[N], double (&gw)[N], bool& finite) {
    const double scale = (MODEL == MODEL_UPPER) ? 0.25 : 1.0;
    double vv[N], dv[N], lam[N];
    int rank[N];
    finite = true;
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {
        finite = finite && d_finite(lam_in[i]);     // before the clamp: fmax drops a NaN
        lam[i] = fmax(lam_in[i], 0.0);
        const double lp = lam[i] * scale;
        const double root = d_sqrt(d_fma(lp, lp, lp));       // sqrt(lp (1 + lp))
        double u = 2.0 * (lp + root);
        double deriv = (lp > 0.0) ? scale * d_rcp(root + TINY) : 0.0;     // dv/dlambda = scale / sqrt(lp(1+lp))
        if (u >= inv_eps - 1.0) {
            const double ilp = d_rcp(1.0 + lp);
            const double d = d_sqrt(lp * ilp);
            const double clamped = d_fma(1.0 + d, inv_eps, -1.0);
            if (clamped < u) {   // v = log((1 + d) / eps):  dv/dlp = 1 / ((1 + d) 2 d (1 + lp)^2)
                u = clamped;
                deriv = scale * d_rcp((1.0 + d) * 2.0 * d) * ilp * ilp;
            }
        }
        vv[i] = d_log1p(u);
        dv[i] = deriv;
    }
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {   // ascending rank, ties broken by index
        int r = 0;
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) r += (vv[j] < vv[i] || (vv[j] == vv[i] && j < i)) ? 1 : 0;
        rank[i] = r;
    }
    double out = 0.0, vbar[N];
    if (metric == METRIC_RIEM) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) out = d_fma(vv[i], vv[i], out);
        out = d_sqrt(out);
        const double inv = (out > 0.0) ? d_rcp(out) : 0.0;
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) vbar[i] = vv[i] * inv;
    } else if (metric == METRIC_FONE) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) { out += vv[i]; vbar[i] = 1.0; }
    } else if (metric == METRIC_FINF) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) { const bool top = rank[i] == N - 1; vbar[i] = top ? 1.0 : 0.0; out += top ? vv[i] : 0.0; }
    } else if (metric == METRIC_FMIN) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) { vbar[i] = 2.0 * rank[i]; out = d_fma(vbar[i], vv[i], out); }
    } else {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) {
            const double wk = w[rank[i]];
            vbar[i] = fmax(wk, 0.0);
            out = d_fma(vbar[i], vv[i], out);
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) gw[k] += (rank[i] == k && wk > 0.0) ? go * vv[i] : 0.0;
        }
    }
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) { phi[i] = go * vbar[i] * dv[i]; philam[i] = phi[i] * lam_in[i]; }

    return out;
}

// S = L^-T M L^-1 for symmetric M (full matrix in, upper triangle i <= j of S out, in m)
template <int N>
SYMPA_HD void sym_congruence_inv_t(const Tri<N, false>& l, double (&m)[N][N]) {
    // W = L^-T M: back substitution down the columns
SYMPA_UNROLL
    for (int c = 0; c < N; ++c) {
SYMPA_UNROLL
        for (int i = N - 1; i >= 0; --i) {
            double t = m[i][c];
SYMPA_UNROLL
            for (int k = i + 1; k < N; ++k) t = d_fma(-l.re[k][i], m[k][c], t);
            m[i][c] = t * l.rdiag[i];
        }
    }
    // S = W L^-1, row r: x_j = (x_j - sum_{k > j} x_k L_kj) / L_jj -- only j >= r is wanted and needs only k > j
SYMPA_UNROLL
    for (int r = 0; r < N; ++r) {
SYMPA_UNROLL
        for (int j = N - 1; j >= r; --j) {
            double t = m[r][j];
SYMPA_UNROLL
            for (int k = j + 1; k < N; ++k) t = d_fma(-m[r][k], l.re[k][j], t);
            m[r][j] = t * l.rdiag[j];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Upper model, n <= 4: everything behind the spectral weights in an order that keeps one lane under 128 doubles (round 6; the
// order of pair_adjoint_gradient_upper, siegel_math_bwd_split.hpp, without a workspace).  The generic tail below holds V, E,
// the full Hbar and K at once (294 registers at n = 4: one wave per SIMD); here
//   1. U = E V in place over E (row by row);  Re G = Re U diag(phi) U^H (symmetric: n (n + 1) / 2 values);
//   2. Ebar = 2 U diag(phi) V^H in place over U (row by row);  Re K = Re V diag(phi lambda) V^H -- V dies.  Never more than
//      E / U / Ebar (one matrix) + V + two symmetric real matrices at once: Hbar itself is never formed;
//   3. Dbar = L1^-T Ebar L2^-1 in place, the factors back from wherever the caller parked them;
//   4. Re planes: g2 = sym(Re Dbar), g1 = -g2;  Im planes: g2 = sym(Im Dbar) - L2^-T Re K L2^-1,  g1 = -sym(Im Dbar) - L1^-T Re G L1^-1.
// Same arithmetic as the generic tail up to the order of the sums.
// ---------------------------------------------------------------------------------------------
template <int N, class Unpark>
SYMPA_HD void upper_adjoint_tail_lean(CMat<N>& e, const CMat<N>& v, const double (&phi)[N], const double (&philam)[N],
                                      Tri<N, false>& l1, Tri<N, false>& l2, Unpark&& unpark, CMat<N>& g1, CMat<N>& g2) {
    // U = E V, row by row over E
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {
        double ur[N], ui[N];
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) {
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int m = 0; m < N; ++m) {
                tr = d_fma(e.re[i][m], v.re[m][k], d_fma(-e.im[i][m], v.im[m][k], tr));
                ti = d_fma(e.re[i][m], v.im[m][k], d_fma(e.im[i][m], v.re[m][k], ti));
            }
            ur[k] = tr; ui[k] = ti;
        }
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) { e.re[i][k] = ur[k]; e.im[i][k] = ui[k]; }
    }
    // Re G = Re U diag(phi) U^H (symmetric), before U is overwritten
    double gg[N][N];
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = i; j < N; ++j) {
            double g = 0.0;
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) g = d_fma(phi[k], d_fma(e.re[i][k], e.re[j][k], e.im[i][k] * e.im[j][k]), g);
            gg[i][j] = g;
            gg[j][i] = g;
        }
    // Ebar = 2 U diag(phi) V^H, row by row over U
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {
        double pr[N], pi[N], fr[N], fi[N];
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) { pr[k] = 2.0 * phi[k] * e.re[i][k]; pi[k] = 2.0 * phi[k] * e.im[i][k]; }
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) {
                tr = d_fma(pr[k], v.re[j][k], d_fma(pi[k], v.im[j][k], tr));
                ti = d_fma(pi[k], v.re[j][k], d_fma(-pr[k], v.im[j][k], ti));
            }
            fr[j] = tr; fi[j] = ti;
        }
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) { e.re[i][j] = fr[j]; e.im[i][j] = fi[j]; }
    }
    // Re K = Re V diag(phi lambda) V^H (symmetric); V dies
    double km[N][N];
SYMPA_UNROLL
    for (int j = 0; j < N; ++j)
SYMPA_UNROLL
        for (int k = j; k < N; ++k) {
            double kr = 0.0;
SYMPA_UNROLL
            for (int i = 0; i < N; ++i) kr = d_fma(philam[i], d_fma(v.re[j][i], v.re[k][i], v.im[j][i] * v.im[k][i]), kr);
            km[j][k] = kr;
            km[k][j] = kr;
        }
    unpark(0, l1);
    solve_lh_left<N, false>(l1, e);
    unpark(1, l2);
    solve_l_right<N, false, true>(l2, e);
    sym_congruence_inv_t<N>(l2, km);
    sym_congruence_inv_t<N>(l1, gg);
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = i; j < N; ++j) {
            const double dr = 0.5 * (e.re[i][j] + e.re[j][i]);
            const double di = 0.5 * (e.im[i][j] + e.im[j][i]);
            g2.re[i][j] = dr;            g2.re[j][i] = dr;
            g1.re[i][j] = -dr;           g1.re[j][i] = -dr;
            const double y2 = di - km[i][j], y1 = -di - gg[i][j];
            g2.im[i][j] = y2;            g2.im[j][i] = y2;
            g1.im[i][j] = y1;            g1.im[j][i] = y1;
        }
}

// ---------------------------------------------------------------------------------------------
// One pair: forward value + gradients.  g1, g2: symmetric matrix gradients w.r.t. Z1, Z2 (both planes);
// gw[k] += d out / d w_k * go for the wsum metric (k = rank of the eigenvalue, ascending).
// ---------------------------------------------------------------------------------------------
// park(k, l) / unpark(k, l): the caller may take the two Cholesky factors out of the registers between the point where E exists
// and the back-substitutions that need them again (the kernel's LDS tile, free between the gather and the scatter): they are the
// 56 registers that kept the n = 4 kernel at one wave per SIMD (294 registers; round 6).  Default: nothing moves.
struct KeepFactors {
    template <class T>
    SYMPA_HD void operator()(int, T&) const {}
};

template <int N, int MODEL, class Park = KeepFactors, class Unpark = KeepFactors>
SYMPA_HD double pair_backward(const CMat<N>& z1, const CMat<N>& z2, int metric, const double* __restrict__ w,
                              double inv_eps, double go, CMat<N>& g1, CMat<N>& g2, double (&gw)[N], int& status,
                              Park&& park = Park(), Unpark&& unpark = Unpark()) {
    constexpr bool CPLX = (MODEL == MODEL_BOUNDED);
    Tri<N, CPLX> l1, l2;
    CMat<N> e;
    bool ok;
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
    park(0, l1);
    solve_right_t<N, CPLX>(l2, e);
    park(1, l2);
    Herm<N> h;
    gram<N>(e, h);
    CMat<N> v;
    bool conv;
    constexpr bool QL_ROUTE = N >= 5;
    if constexpr (QL_ROUTE) conv = herm_eigen_vectors_ql<N>(h, v);
    else conv = herm_eigen_vectors<N>(h, v);
    if constexpr (QL_ROUTE) {
        // Rayleigh quotients lambda_i = v_i^H H v_i = ||E v_i||^2: RELATIVE accuracy for the small eigenvalues, which
        // the QL route needs (column by column: U = E V is never held as a matrix, see below)
SYMPA_UNROLL
        for (int c = 0; c < N; ++c) {
            double acc = 0.0;
SYMPA_UNROLL
            for (int r = 0; r < N; ++r) {
                double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
                for (int k = 0; k < N; ++k) {
                    tr = d_fma(e.re[r][k], v.re[k][c], d_fma(-e.im[r][k], v.im[k][c], tr));
                    ti = d_fma(e.re[r][k], v.im[k][c], d_fma(e.im[r][k], v.re[k][c], ti));
                }
                acc = d_fma(tr, tr, d_fma(ti, ti, acc));
            }
            h.d[c] = acc;
        }
    }

    double phi[N], philam[N];
    bool finite;
    double out = spectral_adjoint<N, MODEL>(h.d, metric, w, inv_eps, go, phi, philam, gw, finite);

    if constexpr (MODEL == MODEL_UPPER && N <= 4) {
        upper_adjoint_tail_lean<N>(e, v, phi, philam, l1, l2, unpark, g1, g2);
        if (!finite) out = __builtin_nan("");
        if (!ok) status |= ST_NOT_PD;
        if (!conv) status |= ST_NO_CONVERGENCE;
        if (!d_finite(out)) status |= ST_NONFINITE;
        return out;
    }

    // closed forms (header comment): no Cholesky adjoint -- the factors only enter through congruences.
    // (Forming U = E V once and Ebar = 2 U diag(phi) V^H, G = U diag(phi) U^H from it saves a product but keeps three
    //  full matrices alive at once: measured SLOWER for the unrolled kernels, n = 8 1.80 ms against 1.44 ms per 262 144
    //  pairs, 1342 against 1102 spilled registers.  The order below frees V, then Hbar, then E as early as possible.)
    CMat<N> hbar, a1, a2;
    herm_from_eig<N>(v, phi, hbar);
    herm_from_eig<N>(v, philam, a2);          // K = V diag(phi lambda) V^H = Hbar H
    CMat<N> ebar;
    cmatmul<N>(e, hbar, 2.0, ebar);           // Ebar = 2 E Hbar
    cmatmul_bh<N>(ebar, e, 0.5, a1);          // G = E Hbar E^H
    unpark(0, l1);
    solve_lh_left<N, CPLX>(l1, ebar);         // Dbar = L1^-H Ebar conj(L2)^-1
    unpark(1, l2);
    solve_l_right<N, CPLX, true>(l2, ebar);
    if (CPLX) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) a2.im[i][j] = -a2.im[i][j];    // conj(K)
    }
    if constexpr (N == 8) {
        // same matrices through the adjoint of the Cholesky factorisation (L_k bar = -2 tril(L_k^-H M), then
        // herm(L^-H Phi(L^H Lbar) L^-1)): more arithmetic, but at n = 8 the register allocator spills less around it
        // (1.44 ms against 1.55 ms per 262 144 pairs; n <= 7: the congruences are 0-3 % faster)
        CMat<N> g = a1, k = a2;
        chol_adjoint<N, CPLX>(l1, g, -2.0, a1);
        chol_adjoint<N, CPLX>(l2, k, -2.0, a2);
    } else {
        neg_congruence<N, CPLX>(l1, a1);      // A1bar = -L1^-H G L1^-1
        neg_congruence<N, CPLX>(l2, a2);      // A2bar = -L2^-H conj(K) L2^-1
    }

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
    if (!finite) out = __builtin_nan("");     // non-finite input: the forward value is NaN like the gradients
    if (!ok) status |= ST_NOT_PD;
    if (!conv) status |= ST_NO_CONVERGENCE;
    if (!d_finite(out)) status |= ST_NONFINITE;
    return out;
}

}  // namespace sympa
