// SPD model, backward and optimiser-side row operations (PARITY UNPINNED like the forward: every line of
// this arithmetic lives in geoopt -- SymmetricPositiveDefinite.{dist, egrad2rgrad, retr, projx}, RiemannianSGD -- which
// is absent from the reference tree; the formulas are restated from the published source and pinned by 50-digit mpmath
// finite differences (tests/golden/spd_n*.npz) and by autograd through the oracle).
//
//   dist(x, y) = sqrt(sum_i log^2(1 + a_i)),   A = L^-1 (y - x) L^-T = V diag(a) V^T,   x = L L^T
//   d dist / d y =  L^-T V diag( log(1 + a_i) / ((1 + a_i) dist) ) V^T L^-1
//   d dist / d x = -L^-T V diag( log(1 + a_i) /              dist ) V^T L^-1         (both symmetric)
// (the second line is -x^-1 Log_x(y) x^-1 / dist, the Euclidean gradient of the affine-invariant distance).
//
//   egrad2rgrad(x, u) = x sym(u) x                                  geoopt SymmetricPositiveDefinite.egrad2rgrad
//   retr(x, u)        = sym(x + u + 1/2 u x^-1 u)                   geoopt SymmetricPositiveDefinite.retr
//   projx(x)          = V |lambda| V^T of sym(x)                    geoopt SymmetricPositiveDefinite.projx
//   RSGD step         : x <- retr(x, -lr egrad2rgrad(x, g + wd x))  geoopt.optim.RiemannianSGD, momentum 0
//
// Runtime n <= 16, one pair (row) per lane, per-lane scratch arrays: functional, not tuned (the forward has the
// sixteen-lanes-per-pair kernel; this is the training path of configs[4]).  Compiled by hipcc and by g++ (hostsim).
#pragma once

#include "spd_math.hpp"

namespace sympa {

struct SpdBwdWork {
    double l[SPD_MAX_N * SPD_MAX_N];   // Cholesky factor of x (lower)
    double a[SPD_MAX_N * SPD_MAX_N];   // A = L^-1 (y - x) L^-T, destroyed by the Jacobi iteration
    double v[SPD_MAX_N * SPD_MAX_N];   // eigenvectors (columns)
    double p[SPD_MAX_N * SPD_MAX_N];   // V diag(.) V^T, then L^-T P L^-1
    double rd[SPD_MAX_N], lam[SPD_MAX_N], f[SPD_MAX_N];
};

// Cyclic Jacobi with eigenvectors on a runtime-n symmetric matrix (full storage, both triangles kept in step).
// a is destroyed, lam = eigenvalues, v = eigenvectors in columns.
SYMPA_HD bool spd_eigh_jacobi(double* a, double* v, double* lam, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
    bool conv = (n <= 1);
    for (int sweep = 0; sweep < 30 && n > 1; ++sweep) {
        double off2 = 0.0, diag2 = 0.0;
        for (int i = 0; i < n; ++i) {
            diag2 += a[i * n + i] * a[i * n + i];
            for (int j = i + 1; j < n; ++j) off2 += a[i * n + j] * a[i * n + j];
        }
        conv = !(off2 > 1e-30 * diag2);
        if (wave_all(conv)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double b = a[p * n + q];
                const double a2 = b * b;
                const double delta = a[q * n + q] - a[p * n + p];
                const double ad = fabs(delta) + 1e-150;
                const double qr = d_rsqrt(d_fma(ad, ad, 4.0 * a2));
                const double c2 = d_fma(0.5 * ad, qr, 0.5);
                const double ic = d_rsqrt(c2);
                const double c = c2 * ic;
                const double cu = copysign(qr, delta) * ic;
                const double s = cu * b;
                const double ua2 = (cu * ic) * a2;
                a[p * n + p] -= ua2;
                a[q * n + q] += ua2;
                a[p * n + q] = 0.0;
                a[q * n + p] = 0.0;
                for (int k = 0; k < n; ++k) {
                    if (k != p && k != q) {
                        const double x = a[k * n + p], y = a[k * n + q];
                        const double nx = c * x - s * y, ny = s * x + c * y;
                        a[k * n + p] = nx; a[p * n + k] = nx;
                        a[k * n + q] = ny; a[q * n + k] = ny;
                    }
                    const double vx = v[k * n + p], vy = v[k * n + q];
                    v[k * n + p] = c * vx - s * vy;
                    v[k * n + q] = s * vx + c * vy;
                }
            }
    }
    for (int i = 0; i < n; ++i) lam[i] = a[i * n + i];
    return conv;
}

// The same decomposition through Householder tridiagonalisation + implicit QL with accumulated transformations
// (EISPACK tred2 / tql2, restated): ~4/3 n^3 + ~3 n^3 operations instead of ~10 sweeps x 2 n^3 of the Jacobi iteration,
// 2.5-3x less work at n = 16.  One matrix per lane with its own (divergent) iteration counts.  a is destroyed.
SYMPA_HD bool spd_eigh_ql(double* a, double* v, double* lam, double* e, int n) {
    // tred2: v <- a; reduce to tridiagonal form, accumulate the orthogonal transformation in v
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[i * n + j] = a[i * n + j];
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += fabs(v[i * n + k]);
            if (!(scale > 0.0)) {
                e[i] = v[i * n + l];
            } else {
                const double iscale = d_rcp(scale);
                for (int k = 0; k <= l; ++k) {
                    v[i * n + k] *= iscale;
                    h += v[i * n + k] * v[i * n + k];
                }
                double f = v[i * n + l];
                double g = -copysign(d_sqrt(h), f);
                e[i] = scale * g;
                h -= f * g;
                v[i * n + l] = f - g;
                f = 0.0;
                const double ih = d_rcp(h);
                for (int j = 0; j <= l; ++j) {
                    v[j * n + i] = v[i * n + j] * ih;
                    g = 0.0;
                    for (int k = 0; k <= j; ++k) g += v[j * n + k] * v[i * n + k];
                    for (int k = j + 1; k <= l; ++k) g += v[k * n + j] * v[i * n + k];
                    e[j] = g * ih;
                    f += e[j] * v[i * n + j];
                }
                const double hh = f * d_rcp(h + h);
                for (int j = 0; j <= l; ++j) {
                    f = v[i * n + j];
                    e[j] = g = e[j] - hh * f;
                    for (int k = 0; k <= j; ++k) v[j * n + k] -= f * e[k] + g * v[i * n + k];
                }
            }
        } else {
            e[i] = v[i * n + l];
        }
        lam[i] = h;
    }
    lam[0] = 0.0;
    e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if (lam[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g = 0.0;
                for (int k = 0; k <= l; ++k) g += v[i * n + k] * v[k * n + j];
                for (int k = 0; k <= l; ++k) v[k * n + j] -= g * v[k * n + i];
            }
        }
        lam[i] = v[i * n + i];
        v[i * n + i] = 1.0;
        for (int j = 0; j <= l; ++j) { v[j * n + i] = 0.0; v[i * n + j] = 0.0; }
    }
    // tql2
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    bool ok = true;
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        int m;
        do {
            for (m = l; m < n - 1; ++m) {
                if (ql_negligible(e[m] * e[m], lam[m], lam[m + 1])) break;
            }
            if (m != l) {
                if (iter++ == 60) { ok = false; break; }
                double g = (lam[l + 1] - lam[l]) * 0.5 * d_rcp(e[l]);
                double r = d_sqrt(d_fma(g, g, 1.0));
                g = lam[m] - lam[l] + e[l] * d_rcp(g + copysign(r, g));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = d_sqrt(d_fma(f, f, g * g));
                    e[i + 1] = r;
                    if (!(r > 0.0)) {
                        lam[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    const double ir = d_rcp(r);
                    s = f * ir;
                    c = g * ir;
                    g = lam[i + 1] - p;
                    r = d_fma(lam[i] - g, s, 2.0 * c * b);
                    p = s * r;
                    lam[i + 1] = g + p;
                    g = d_fma(c, r, -b);
                    for (int k = 0; k < n; ++k) {
                        f = v[k * n + i + 1];
                        v[k * n + i + 1] = d_fma(s, v[k * n + i], c * f);
                        v[k * n + i] = d_fma(c, v[k * n + i], -s * f);
                    }
                }
                if (!(r > 0.0) && i >= l) continue;
                lam[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    return ok;
}

// Cholesky x = L L^T into w.l / w.rd (lower); returns "x is positive definite".
SYMPA_HD bool spd_cholesky(const double* __restrict__ px, int n, double* l, double* rd) {
    bool ok = true;
    for (int j = 0; j < n; ++j) {
        double s = sym_at(px, n, j, j);
        for (int k = 0; k < j; ++k) s -= l[j * n + k] * l[j * n + k];
        ok = ok && (s > 0.0);
        const double r = d_rsqrt(s);
        rd[j] = r;
        l[j * n + j] = s * r;
        for (int i = j + 1; i < n; ++i) {
            double t = sym_at(px, n, i, j);
            for (int k = 0; k < j; ++k) t -= l[i * n + k] * l[j * n + k];
            l[i * n + j] = t * r;
        }
    }
    return ok;
}

// m <- L^-1 m L^-T  (m full n x n)
SYMPA_HD void spd_congruence_inv(double* m, const double* l, const double* rd, int n) {
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < n; ++i) {
            double t = m[i * n + c];
            for (int k = 0; k < i; ++k) t -= l[i * n + k] * m[k * n + c];
            m[i * n + c] = t * rd[i];
        }
    for (int r = 0; r < n; ++r)
        for (int j = 0; j < n; ++j) {
            double t = m[r * n + j];
            for (int k = 0; k < j; ++k) t -= m[r * n + k] * l[j * n + k];
            m[r * n + j] = t * rd[j];
        }
}

// m <- L^-T m L^-1  (m full n x n): back substitutions
SYMPA_HD void spd_congruence_inv_t(double* m, const double* l, const double* rd, int n) {
    for (int c = 0; c < n; ++c)                       // L^T t = m[:, c]
        for (int i = n - 1; i >= 0; --i) {
            double t = m[i * n + c];
            for (int k = i + 1; k < n; ++k) t -= l[k * n + i] * m[k * n + c];
            m[i * n + c] = t * rd[i];
        }
    for (int r = 0; r < n; ++r)                       // t L = m[r, :]
        for (int j = n - 1; j >= 0; --j) {
            double t = m[r * n + j];
            for (int k = j + 1; k < n; ++k) t -= m[r * n + k] * l[k * n + j];
            m[r * n + j] = t * rd[j];
        }
}

// One pair: returns dist(x, y); gx, gy ([n, n], full symmetric) = d dist / d x, d dist / d y  (multiply by the incoming
// gradient outside).  dist = 0 (x = y) has the zero subgradient.
SYMPA_HD double spd_pair_backward(SpdBwdWork& w, const double* __restrict__ px, const double* __restrict__ py, int n,
                                  double* __restrict__ gx, double* __restrict__ gy, int& status) {
    bool ok = spd_cholesky(px, n, w.l, w.rd);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) w.a[i * n + j] = sym_at(py, n, i, j) - sym_at(px, n, i, j);
    spd_congruence_inv(w.a, w.l, w.rd, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) { const double t = 0.5 * (w.a[i * n + j] + w.a[j * n + i]); w.a[i * n + j] = t; w.a[j * n + i] = t; }
    // tridiagonal QL (eigenvalues to eps ||A||), then Rayleigh quotients lam_i = v_i^T A v_i with the saved matrix for the
    // eigenvalues near zero (x ~ y in some directions), where the logarithm's relative accuracy matters
    for (int k = 0; k < n * n; ++k) w.p[k] = w.a[k];
    const bool conv = spd_eigh_ql(w.a, w.v, w.lam, w.f, n);
    for (int c = 0; c < n; ++c) {
        double acc2 = 0.0;
        for (int i = 0; i < n; ++i) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.p[i * n + k] * w.v[k * n + c];
            acc2 += w.v[i * n + c] * t;
        }
        w.lam[c] = acc2;
    }
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        ok = ok && (w.lam[i] > -1.0);
        w.f[i] = d_log1p_signed(w.lam[i]);
        acc = d_fma(w.f[i], w.f[i], acc);
    }
    const double dist = d_sqrt(acc);
    const double inv = (dist > 0.0) ? d_rcp(dist) : 0.0;
    // gy = L^-T V diag(f / (1 + lam)) V^T L^-1 / dist
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.v[i * n + k] * (w.f[k] * d_rcp(1.0 + w.lam[k])) * w.v[j * n + k];
            w.p[i * n + j] = t * inv;
            w.p[j * n + i] = t * inv;
        }
    spd_congruence_inv_t(w.p, w.l, w.rd, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) gy[i * n + j] = 0.5 * (w.p[i * n + j] + w.p[j * n + i]);
    // gx = -L^-T V diag(f) V^T L^-1 / dist
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.v[i * n + k] * w.f[k] * w.v[j * n + k];
            w.p[i * n + j] = -t * inv;
            w.p[j * n + i] = -t * inv;
        }
    spd_congruence_inv_t(w.p, w.l, w.rd, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) gx[i * n + j] = 0.5 * (w.p[i * n + j] + w.p[j * n + i]);
    if (!ok) status |= ST_NOT_PD;
    if (!conv) status |= ST_NO_CONVERGENCE;
    if (!d_finite(dist)) status |= ST_NONFINITE;
    return dist;
}

// ---- rows of the table -------------------------------------------------------------------------------------------
struct SpdRowWork {
    double x[SPD_MAX_N * SPD_MAX_N], u[SPD_MAX_N * SPD_MAX_N], t[SPD_MAX_N * SPD_MAX_N], l[SPD_MAX_N * SPD_MAX_N];
    double rd[SPD_MAX_N], lam[SPD_MAX_N];
};

// out = x sym(u) x
SYMPA_HD void spd_row_egrad2rgrad(SpdRowWork& w, const double* __restrict__ px, const double* __restrict__ pu, int n,
                                  double* __restrict__ out) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            w.x[i * n + j] = 0.5 * (px[i * n + j] + px[j * n + i]);
            w.u[i * n + j] = 0.5 * (pu[i * n + j] + pu[j * n + i]);
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.x[i * n + k] * w.u[k * n + j];
            w.t[i * n + j] = t;
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.t[i * n + k] * w.x[k * n + j];
            out[i * n + j] = t;
        }
}

// out = V |lambda| V^T of sym(x); returns true when some eigenvalue was negative (the row moved)
SYMPA_HD bool spd_row_projx(SpdRowWork& w, const double* __restrict__ px, int n, double* __restrict__ out, int& status) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) w.x[i * n + j] = 0.5 * (px[i * n + j] + px[j * n + i]);
    for (int k = 0; k < n * n; ++k) w.t[k] = w.x[k];
    const bool conv = spd_eigh_jacobi(w.t, w.u, w.lam, n);
    bool moved = false;
    for (int i = 0; i < n; ++i) moved = moved || (w.lam[i] < 0.0);
    if (!conv) status |= ST_NO_CONVERGENCE;
    if (!moved) {                     // already inside: only the symmetrisation
        for (int k = 0; k < n * n; ++k) out[k] = w.x[k];
        return false;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.u[i * n + k] * fabs(w.lam[k]) * w.u[j * n + k];
            out[i * n + j] = t;
        }
    return true;
}

// x <- retr(x, -lr * x sym(g + wd x) x) in place;  retr(x, u) = sym(x + u + 1/2 u x^-1 u) = sym(x + u + 1/2 W^T W),
// W = L^-1 u.  `coef` scales the gradient (gradient clipping).
SYMPA_HD void spd_row_rsgd(SpdRowWork& w, double* __restrict__ px, const double* __restrict__ pg, int n, double lr,
                           double wd, double coef, int& status) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            w.x[i * n + j] = 0.5 * (px[i * n + j] + px[j * n + i]);
            w.u[i * n + j] = 0.5 * coef * (pg[i * n + j] + pg[j * n + i]);
        }
    if (wd != 0.0)
        for (int k = 0; k < n * n; ++k) w.u[k] = d_fma(wd, w.x[k], w.u[k]);
    // u <- -lr x u x
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.x[i * n + k] * w.u[k * n + j];
            w.t[i * n + j] = t;
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.t[i * n + k] * w.x[k * n + j];
            w.u[i * n + j] = -lr * t;
        }
    const bool ok = spd_cholesky(w.x, n, w.l, w.rd);
    // W = L^-1 u  (columns)
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < n; ++i) {
            double t = w.u[i * n + c];
            for (int k = 0; k < i; ++k) t -= w.l[i * n + k] * w.t[k * n + c];
            w.t[i * n + c] = t * w.rd[i];
        }
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += w.t[k * n + i] * w.t[k * n + j];
            const double val = w.x[i * n + j] + 0.5 * (w.u[i * n + j] + w.u[j * n + i]) + 0.5 * t;
            px[i * n + j] = val;
            px[j * n + i] = val;
        }
    if (!ok) status |= ST_NOT_PD;
}

}  // namespace sympa
