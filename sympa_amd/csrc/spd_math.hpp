// SPD model (BASELINE.json configs[4]): affine-invariant distance between symmetric positive
// definite matrices.  In the reference the `spd` model is geoopt.manifolds.SymmetricPositiveDefinite()
// (sympa/embeddings.py:6,70-72,142); ALL of its arithmetic lives in geoopt, an un-vendored dependency absent
// from the reference tree, so there is nothing of the reference to pin against: PARITY UNPINNED.  The oracle
// restates geoopt's published AIM formula  dist(x, y) = || log(x^-1/2 y x^-1/2) ||_F  (eigh-based inverse square
// root and logm, geoopt/manifolds/symmetric_positive_definite.py, geoopt/linalg/batch_linalg.py).
//
// Evaluated here as  sqrt(sum_i log^2(1 + mu_i)),  mu = eigenvalues of  L^-1 (Y - X) L^-T,  X = L L^T:
// one Cholesky, two triangular solves, Householder tridiagonalisation, PWK QL -- and, as for the Siegel models,
// the exact difference Y - X keeps full relative accuracy for nearby points (the init distribution is I + 1e-3).
// Runtime-n version with per-lane scratch arrays (n <= 16); compiled by hipcc and by g++ (tests/hostsim).
// A register / MFMA-tiled n = 16 kernel is the next step for this model.
#pragma once

#include "siegel_math.hpp"
#include "siegel_math_generic.hpp"

namespace sympa {

constexpr int SPD_MAX_N = 16;

struct SpdWork {
    double a[SPD_MAX_N * SPD_MAX_N];   // Y - X, then L^-1 (Y-X) L^-T, then the Householder workspace
    double l[SPD_MAX_N * SPD_MAX_N];   // Cholesky factor of X (lower)
    double rd[SPD_MAX_N];              // 1 / diag(L)
    double d[SPD_MAX_N], e2[SPD_MAX_N], v[SPD_MAX_N], p[SPD_MAX_N];
};

// log(1 + u) for u > -1 (the Siegel epilogue only needs u >= 0)
SYMPA_HD double d_log1p_signed(double u) {
    const double x = 1.0 + u;
    double m = d_frexp_mant(x);
    int k = d_frexp_exp(x);
    const bool lowhalf = m < 0.70710678118654752440;
    m = lowhalf ? 2.0 * m : m;
    k = lowhalf ? k - 1 : k;
    const bool small = (u < 0.4) && (u > -0.29);
    const double f = small ? u : m - 1.0;
    const double kd = small ? 0.0 : (double)k;
    const double s = f * d_rcp(2.0 + f);
    const double z = s * s;
    double r = 1.479819860511658591e-01;
    r = d_fma(r, z, 1.531383769920937332e-01);
    r = d_fma(r, z, 1.818357216161805012e-01);
    r = d_fma(r, z, 2.222219843214978396e-01);
    r = d_fma(r, z, 2.857142874366239149e-01);
    r = d_fma(r, z, 3.999999999940941908e-01);
    r = d_fma(r, z, 6.666666666666735130e-01);
    r = r * z;
    const double l = d_fma(-s, f - r, f);
    return d_fma(kd, 6.93147180559945286227e-01, l);
}

// Eigenvalues of the symmetric tridiagonal (d, e2) with runtime n, PWK QL (dsterf), in place in d.
SYMPA_HD bool tridiag_ql_runtime(double* d, double* e2, int n) {
    constexpr double TOL = 1.3e-32;
    int l = 0;
    bool done = (n <= 1);
    for (int iter = 0; iter < 40 * SPD_MAX_N && n > 1; ++iter) {
        for (int i = 0; i < n - 1; ++i)
            if (e2[i] <= TOL * fabs(d[i] * d[i + 1]) + 1e-290) e2[i] = 0.0;
        while (l < n - 1 && e2[l] == 0.0) ++l;
        done = l >= n - 1;
        if (wave_all(done)) break;
        if (!done) {
            int m = l;
            while (m < n - 1 && e2[m] != 0.0) ++m;
            const double el = e2[l];
            const double irte = d_rsqrt(el);
            const double rte = el * irte;
            const double sg = 0.5 * (d[l + 1] - d[l]) * irte;
            const double rr = d_sqrt(d_fma(sg, sg, 1.0));
            const double sigma = d[l] - rte * d_rcp(sg + copysign(rr, sg));
            double c = 1.0, sn = 0.0, gamma = d[m] - sigma, p = gamma * gamma;
            for (int i = m - 1; i >= l; --i) {
                const double bb = e2[i];
                const double r = p + bb;
                if (i != m - 1) e2[i + 1] = sn * r;
                const double oldc = c;
                const double ir = d_rcp(r);
                c = p * ir;
                sn = bb * ir;
                const double oldgam = gamma;
                const double alpha = d[i];
                gamma = d_fma(c, alpha - sigma, -sn * oldgam);
                d[i + 1] = oldgam + (alpha - gamma);
                p = (c != 0.0) ? gamma * gamma * d_rcp(c) : oldc * bb;
            }
            e2[l] = sn * p;
            d[l] = sigma + gamma;
        }
    }
    return done;
}

// Householder tridiagonalisation of a real symmetric S x S matrix held PACKED by one lane: element (i, j), i >= j, is
// a[i (i + 1) / 2 + j]; fully unrolled, every index a constant, so the 36 doubles of an 8 x 8 block live in registers.
// Out: d[0 .. S), e2[0 .. S - 1) (squares of the off-diagonal).  The lanes-per-pair kernels (spd_coop.hpp) hand the
// trailing block of every pair to this routine: a step here serves 64 pairs per wave instruction, a step in the
// row-per-lane layout four.
template <int S>
SYMPA_HD void tridiag_packed(double (&a)[S * (S + 1) / 2], double* __restrict__ d, double* __restrict__ e2) {
#define SYMPA_AP(i, j) a[((i) >= (j)) ? ((i) * ((i) + 1) / 2 + (j)) : ((j) * ((j) + 1) / 2 + (i))]
SYMPA_UNROLL
    for (int k = 0; k < S - 2; ++k) {
        double s2 = 0.0;
SYMPA_UNROLL
        for (int i = k + 2; i < S; ++i) s2 = d_fma(SYMPA_AP(i, k), SYMPA_AP(i, k), s2);
        const double x0 = SYMPA_AP(k + 1, k);
        const double n2 = d_fma(x0, x0, s2);
        d[k] = SYMPA_AP(k, k);
        e2[k] = n2;
        const double nx = d_sqrt(n2);
        const double v0 = x0 + copysign(nx, x0);
        const double den = d_fma(v0, v0, s2);
        const double beta = (den > 0.0) ? 2.0 * d_rcp(den) : 0.0;
        double v[S], p[S];
        v[k + 1] = v0;
SYMPA_UNROLL
        for (int i = k + 2; i < S; ++i) v[i] = SYMPA_AP(i, k);
        double kk = 0.0;
SYMPA_UNROLL
        for (int i = k + 1; i < S; ++i) {
            double t = 0.0;
SYMPA_UNROLL
            for (int j = k + 1; j < S; ++j) t = d_fma(SYMPA_AP(i, j), v[j], t);
            p[i] = beta * t;
            kk = d_fma(v[i], p[i], kk);
        }
        kk *= 0.5 * beta;
SYMPA_UNROLL
        for (int i = k + 1; i < S; ++i) p[i] = d_fma(-kk, v[i], p[i]);
SYMPA_UNROLL
        for (int i = k + 1; i < S; ++i) {
SYMPA_UNROLL
            for (int j = k + 1; j <= i; ++j) SYMPA_AP(i, j) = d_fma(-v[i], p[j], d_fma(-p[i], v[j], SYMPA_AP(i, j)));
        }
    }
    if (S >= 2) {
        d[S - 2] = SYMPA_AP(S - 2, S - 2);
        e2[S - 2] = SYMPA_AP(S - 1, S - 2) * SYMPA_AP(S - 1, S - 2);
    }
    d[S - 1] = SYMPA_AP(S - 1, S - 1);
#undef SYMPA_AP
}

// px, py: [n, n] fp64 row-major symmetric (upper triangle read).  Returns the AIM distance.
SYMPA_HD double spd_pair_distance(SpdWork& w, const double* __restrict__ px, const double* __restrict__ py, int n,
                                  int& status) {
    // Cholesky X = L L^T
    bool ok = true;
    for (int j = 0; j < n; ++j) {
        double s = sym_at(px, n, j, j);
        for (int k = 0; k < j; ++k) s -= w.l[j * n + k] * w.l[j * n + k];
        ok = ok && (s > 0.0);
        const double r = d_rsqrt(s);
        w.rd[j] = r;
        for (int i = j + 1; i < n; ++i) {
            double t = sym_at(px, n, i, j);
            for (int k = 0; k < j; ++k) t -= w.l[i * n + k] * w.l[j * n + k];
            w.l[i * n + j] = t * r;
        }
    }
    // A = Y - X (full), A <- L^-1 A, A <- A L^-T
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) w.a[i * n + j] = sym_at(py, n, i, j) - sym_at(px, n, i, j);
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < n; ++i) {
            double t = w.a[i * n + c];
            for (int k = 0; k < i; ++k) t -= w.l[i * n + k] * w.a[k * n + c];
            w.a[i * n + c] = t * w.rd[i];
        }
    for (int r = 0; r < n; ++r)
        for (int j = 0; j < n; ++j) {
            double t = w.a[r * n + j];
            for (int k = 0; k < j; ++k) t -= w.a[r * n + k] * w.l[j * n + k];
            w.a[r * n + j] = t * w.rd[j];
        }
    // Y itself must be positive definite as well: mu_i > -1 is checked below through the logarithm's argument
    // Householder tridiagonalisation (lower triangle of A is the working copy; symmetrised on the fly)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) { const double t = 0.5 * (w.a[i * n + j] + w.a[j * n + i]); w.a[i * n + j] = t; w.a[j * n + i] = t; }
    for (int k = 0; k < n - 2; ++k) {
        double sig2 = 0.0;
        for (int i = k + 2; i < n; ++i) sig2 += w.a[i * n + k] * w.a[i * n + k];
        const double x0 = w.a[(k + 1) * n + k];
        const double n2 = x0 * x0 + sig2;
        w.d[k] = w.a[k * n + k];
        w.e2[k] = n2;
        if (!(sig2 > 0.0)) continue;
        const double nx = d_sqrt(n2);
        w.v[k + 1] = x0 + copysign(nx, x0);
        for (int i = k + 2; i < n; ++i) w.v[i] = w.a[i * n + k];
        const double beta = 2.0 * d_rcp(w.v[k + 1] * w.v[k + 1] + sig2);
        double kk = 0.0;
        for (int i = k + 1; i < n; ++i) {
            double t = 0.0;
            for (int j = k + 1; j < n; ++j) t += w.a[i * n + j] * w.v[j];
            w.p[i] = beta * t;
            kk += w.v[i] * w.p[i];
        }
        kk *= 0.5 * beta;
        for (int i = k + 1; i < n; ++i) w.p[i] -= kk * w.v[i];
        for (int i = k + 1; i < n; ++i)
            for (int j = k + 1; j < n; ++j) w.a[i * n + j] -= w.v[i] * w.p[j] + w.p[i] * w.v[j];
    }
    if (n >= 2) {
        w.d[n - 2] = w.a[(n - 2) * n + (n - 2)];
        w.e2[n - 2] = w.a[(n - 1) * n + (n - 2)] * w.a[(n - 1) * n + (n - 2)];
    }
    w.d[n - 1] = w.a[(n - 1) * n + (n - 1)];
    const bool conv = tridiag_ql_runtime(w.d, w.e2, n);
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        ok = ok && (w.d[i] > -1.0);
        const double lg = d_log1p_signed(w.d[i]);
        acc = d_fma(lg, lg, acc);
    }
    const double out = d_sqrt(acc);
    if (!ok) status |= ST_NOT_PD;
    if (!conv) status |= ST_NO_CONVERGENCE;
    if (!(out == out) || !(fabs(out) <= 1.79e308)) status |= ST_NONFINITE;
    return out;
}

}  // namespace sympa
