// Per-row arithmetic of the optimiser-side manifold operations over the embedding table,
// compiled by hipcc (kernels) and by g++ (tests/hostsim), like siegel_math.hpp.
//   upper  : egrad2rgrad = Y G Y on both planes                    sympa/manifolds/upper_half.py:25-40
//            projx = symmetrise, then clamp the eigenvalues of Y at eps, untouched when all > eps
//                                                                   upper_half.py:42-66, csym_math.py:252-278
//   bounded: egrad2rgrad = A G A,  A = I - conj(Z) Z               sympa/manifolds/bounded_domain.py:41-53,163-170
//            projx = symmetrise, then clamp the Takagi values of Z at 1 - eps, untouched when all < 1 - eps
//                    (intended behaviour of bounded_domain.py:55-84; the in-tree call is broken)
//   retr   = projx(x + u)                                           sympa/manifolds/siegel_manifold.py:74-87
//   RSGD step (geoopt.optim.RiemannianSGD with momentum 0, the optimiser train.py:66-68 builds):
//            x <- retr(x, -lr * egrad2rgrad(x, grad + weight_decay * x))
#pragma once

#include "siegel_math.hpp"
#include "siegel_math_bwd.hpp"

namespace sympa {

// Real symmetric Jacobi with eigenvectors: a = V diag(d) V^T.  a: full symmetric input (upper triangle used).
template <int N>
SYMPA_HD bool sym_eigen_vectors(double (&a)[N][N], double (&d)[N], double (&v)[N][N]) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {
        d[i] = a[i][i];
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    }
    if (N == 1) return true;
    bool conv = false;
    for (int sweep = 0; sweep < JACOBI_MAX_SWEEPS; ++sweep) {
        double off2 = 0.0, diag2 = 0.0;
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) {
            diag2 = d_fma(d[i], d[i], diag2);
SYMPA_UNROLL
            for (int j = i + 1; j < N; ++j) off2 = d_fma(a[i][j], a[i][j], off2);
        }
        conv = !(off2 > 1e-30 * diag2);
        if (wave_all(conv)) break;
SYMPA_UNROLL
        for (int p = 0; p < N - 1; ++p) {
SYMPA_UNROLL
            for (int q = p + 1; q < N; ++q) {
                const double b = a[p][q];
                const double a2 = b * b;
                const double delta = d[q] - d[p];
                const double ad = fabs(delta) + 1e-150;
                const double qr = d_rsqrt(d_fma(ad, ad, 4.0 * a2));
                const double c2 = d_fma(0.5 * ad, qr, 0.5);
                const double ic = d_rsqrt(c2);
                const double c = c2 * ic;
                const double cu = copysign(qr, delta) * ic;
                const double s = cu * b;                 // sin, signed
                const double ua2 = (cu * ic) * a2;
                d[p] -= ua2;
                d[q] += ua2;
                a[p][q] = 0.0;
SYMPA_UNROLL
                for (int k = 0; k < N; ++k) {
                    if (k == p || k == q) continue;
                    const double x = (k < p) ? a[k][p] : a[p][k];
                    const double y = (k < q) ? a[k][q] : a[q][k];
                    const double nx = d_fma(-s, y, c * x);
                    const double ny = d_fma(s, x, c * y);
                    if (k < p) a[k][p] = nx; else a[p][k] = nx;
                    if (k < q) a[k][q] = ny; else a[q][k] = ny;
                }
SYMPA_UNROLL
                for (int k = 0; k < N; ++k) {
                    const double x = v[k][p], y = v[k][q];
                    v[k][p] = d_fma(-s, y, c * x);
                    v[k][q] = d_fma(s, x, c * y);
                }
            }
        }
    }
    return conv;
}

template <int N>
SYMPA_HD void symmetrise(CMat<N>& z) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = i + 1; j < N; ++j) {
            const double r = 0.5 * (z.re[i][j] + z.re[j][i]);
            const double m = 0.5 * (z.im[i][j] + z.im[j][i]);
            z.re[i][j] = r; z.re[j][i] = r;
            z.im[i][j] = m; z.im[j][i] = m;
        }
}

// real n x n products  out = a * b
template <int N>
SYMPA_HD void rmatmul(const double (&a)[N][N], const double (&b)[N][N], double (&out)[N][N]) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            double t = 0.0;
SYMPA_UNROLL
            for (int k = 0; k < N; ++k) t = d_fma(a[i][k], b[k][j], t);
            out[i][j] = t;
        }
}

// ---- egrad2rgrad --------------------------------------------------------------------------------
template <int N, int MODEL>
SYMPA_HD void egrad2rgrad(const CMat<N>& z, const CMat<N>& u, CMat<N>& out) {
    if (MODEL == MODEL_UPPER) {
        double t[N][N];
        rmatmul<N>(z.im, u.re, t);
        rmatmul<N>(t, z.im, out.re);
        rmatmul<N>(z.im, u.im, t);
        rmatmul<N>(t, z.im, out.im);
    } else {
        CMat<N> a, zc, t;
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) { zc.re[i][j] = z.re[i][j]; zc.im[i][j] = -z.im[i][j]; }
        cmatmul<N>(zc, z, -1.0, a);              // -conj(Z) Z
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) a.re[i][i] += 1.0;
        cmatmul<N>(a, u, 1.0, t);
        cmatmul<N>(t, a, 1.0, out);
    }
}

// ---- projx --------------------------------------------------------------------------------------
// Returns true when the point had to be moved (counts towards manifold.projected_points).
template <int N, int MODEL>
SYMPA_HD bool projx(CMat<N>& z, double eps, int& status) {
    symmetrise<N>(z);
    // Cheap certificate first: the row is inside the eps-interior iff a Cholesky factorisation exists (upper: Im z - eps I
    // positive definite <=> every eigenvalue > eps; bounded: I - Z Z^H / (1 - eps)^2 positive definite <=> every Takagi value
    // < 1 - eps).  When it holds for all 64 rows of the wave -- every step of a converging run -- the eigendecomposition
    // (~25 x the instructions) is skipped and the rows stay untouched, exactly what the reference's mask does
    // (upper_half.py:42-66, bounded_domain.py:55-84).  The certificate is taken a hair (1e-9) inside the boundary, so a row
    // within rounding of it still gets the exact path and the reference's own comparison.
    {
        constexpr double HAIR = 1e-9;
        bool certain;
        if (MODEL == MODEL_UPPER) {
            double a[N][N];
            const double shift = eps * (1.0 + HAIR);
SYMPA_UNROLL
            for (int i = 0; i < N; ++i)
SYMPA_UNROLL
                for (int j = 0; j < N; ++j) a[i][j] = (i == j) ? z.im[i][j] - shift : z.im[i][j];
            Tri<N, false> l;
            certain = chol_real<N>(a, l);
        } else {
            CMat<N> ws;
            const double inv = 1.0 / ((1.0 - eps) * (1.0 - HAIR));
SYMPA_UNROLL
            for (int i = 0; i < N; ++i)
SYMPA_UNROLL
                for (int j = 0; j < N; ++j) { ws.re[i][j] = z.re[i][j] * inv; ws.im[i][j] = z.im[i][j] * inv; }
            Tri<N, true> c;
            certain = chol_id_minus_wwh<N>(ws, c);
        }
        if (wave_all(certain)) return false;
    }
    if (MODEL == MODEL_UPPER) {
        double a[N][N], d[N], v[N][N];
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) a[i][j] = z.im[i][j];
        if (!sym_eigen_vectors<N>(a, d, v)) status |= ST_NO_CONVERGENCE;
        bool inside = true;
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) inside = inside && (d[i] > eps);
        if (inside) return false;
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) {
                double t = 0.0;
SYMPA_UNROLL
                for (int k = 0; k < N; ++k) t = d_fma(v[i][k] * fmax(d[k], eps), v[j][k], t);
                z.im[i][j] = t;
            }
        return true;
    } else {
        // Takagi values of Z = sqrt(eig(Z^H Z)); clamp sigma_k > 1 - eps:
        //   Z~ = Z - sum_k (sigma_k - (1 - eps)) e^{i theta_k} conj(u_k) u_k^H,   e^{i theta_k} = u_k^T Z u_k / sigma_k
        Herm<N> h;
        gram<N>(z, h);
        CMat<N> v;
        if (!herm_eigen_vectors<N>(h, v)) status |= ST_NO_CONVERGENCE;
        const double lim = 1.0 - eps;
        bool inside = true;
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) inside = inside && (h.d[k] < lim * lim);
        if (inside) return false;
        CMat<N> zu;
        cmatmul<N>(z, v, 1.0, zu);               // columns Z u_k
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) {
            const double sig = d_sqrt(fmax(h.d[k], 0.0));
            if (!(sig > lim)) continue;
            double pr = 0.0, pi = 0.0;            // u_k^T (Z u_k)
SYMPA_UNROLL
            for (int i = 0; i < N; ++i) {
                pr = d_fma(v.re[i][k], zu.re[i][k], pr);
                pr = d_fma(-v.im[i][k], zu.im[i][k], pr);
                pi = d_fma(v.re[i][k], zu.im[i][k], pi);
                pi = d_fma(v.im[i][k], zu.re[i][k], pi);
            }
            const double f = (sig - lim) * d_rcp(sig);          // (sigma - lim) e^{i theta} = f * p,  p = sigma e^{i theta}
            const double fr = f * pr, fi = f * pi;
SYMPA_UNROLL
            for (int i = 0; i < N; ++i)
SYMPA_UNROLL
                for (int j = 0; j < N; ++j) {
                    // conj(u_ik) conj(u_jk)   (conj(u_k) u_k^H)_ij
                    const double cr = v.re[i][k] * v.re[j][k] - v.im[i][k] * v.im[j][k];
                    const double ci = -(v.re[i][k] * v.im[j][k] + v.im[i][k] * v.re[j][k]);
                    z.re[i][j] -= fr * cr - fi * ci;
                    z.im[i][j] -= fr * ci + fi * cr;
                }
        }
        symmetrise<N>(z);
        return true;
    }
}

// ---- one RSGD step on one row --------------------------------------------------------------------
template <int N, int MODEL>
SYMPA_HD bool rsgd_row(CMat<N>& z, const CMat<N>& grad, double lr, double weight_decay, double eps, int& status) {
    CMat<N> g, r;
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            g.re[i][j] = d_fma(weight_decay, z.re[i][j], grad.re[i][j]);
            g.im[i][j] = d_fma(weight_decay, z.im[i][j], grad.im[i][j]);
        }
    egrad2rgrad<N, MODEL>(z, g, r);
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            z.re[i][j] = d_fma(-lr, r.re[i][j], z.re[i][j]);
            z.im[i][j] = d_fma(-lr, r.im[i][j], z.im[i][j]);
        }
    return projx<N, MODEL>(z, eps, status);
}

// full (non-symmetric) load / store of a [2,n,n] row
// ---- inner(z, u, u): squared Riemannian norm of a tangent vector -----------------------------------------------
//   upper  : Re tr[ y^-1 u y^-1 conj(u) ] = Re tr[ Q conj(Q) ],  Q = L^-1 u L^-T,  y = L L^T        upper_half.py:68-91
//   bounded: Re tr[ (I - conj(z) z)^-1 u (I - z conj(z))^-1 conj(u) ] = Re tr[ conj(Q) Q ],  Q = C^-1 conj(u) C^-T,
//            I - z z^H = C C^H                                                                 bounded_domain.py:86-116
// Re tr[Q conj(Q)] = sum_ij (Re q_ij Re q_ji + Im q_ij Im q_ji): the Frobenius norm when u is symmetric (upper: Y G Y
// is), but the bounded Riemannian gradient A G A (A Hermitian, not real) is NOT symmetric.  Used by RiemannianAdam's second moment
// (geoopt Manifold.component_inner = inner broadcast over the point).
template <int N, int MODEL>
SYMPA_HD double tangent_sqnorm(const CMat<N>& z, const CMat<N>& u, int& status) {
    CMat<N> e;
    bool ok;
    if (MODEL == MODEL_UPPER) {
        Tri<N, false> l;
        ok = chol_real<N>(z.im, l);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) { e.re[i][j] = u.re[i][j]; e.im[i][j] = u.im[i][j]; }
        solve_left<N, false>(l, e);
        solve_right_t<N, false>(l, e);
    } else {
        Tri<N, true> c;
        ok = chol_id_minus_wwh<N>(z, c);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i)
SYMPA_UNROLL
            for (int j = 0; j < N; ++j) { e.re[i][j] = u.re[i][j]; e.im[i][j] = -u.im[i][j]; }
        solve_left<N, true>(c, e);
        solve_right_t<N, true>(c, e);
    }
    double acc = 0.0;
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) acc = d_fma(e.re[i][j], e.re[j][i], d_fma(e.im[i][j], e.im[j][i], acc));
    if (!ok) status |= ST_NOT_PD;
    return acc;
}

template <int N>
SYMPA_HD void load_full(const double* __restrict__ p, CMat<N>& z) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) { z.re[i][j] = p[i * N + j]; z.im[i][j] = p[N * N + i * N + j]; }
}
template <int N>
SYMPA_HD void store_full(double* __restrict__ p, const CMat<N>& z) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) { p[i * N + j] = z.re[i][j]; p[N * N + i * N + j] = z.im[i][j]; }
}

}  // namespace sympa
