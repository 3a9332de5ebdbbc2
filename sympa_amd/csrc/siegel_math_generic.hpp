// Runtime-n fallback of the forward arithmetic for dims > SYMPA_MAX_DIMS (up to GENERIC_MAX_N = 16).
// Same algorithm as siegel_math.hpp (Cholesky, triangular solves, Gram matrix, cyclic Hermitian Jacobi, log1p
// epilogue, metric), but with runtime loops over per-lane arrays, which the compiler places in scratch memory:
// correct for any n, an order of magnitude slower per flop than the register-resident specialisations.  It
// exists so that every `--dims` the reference accepts runs on the GPU; compiled by g++ for tests/hostsim too.
#pragma once

#include "siegel_math.hpp"

namespace sympa {

constexpr int GENERIC_MAX_N = 16;

struct GenericWork {
    double er[GENERIC_MAX_N * GENERIC_MAX_N], ei[GENERIC_MAX_N * GENERIC_MAX_N];   // E (n x n complex)
    double hr[GENERIC_MAX_N * GENERIC_MAX_N], hi[GENERIC_MAX_N * GENERIC_MAX_N];   // H / factor workspace
    double lr[GENERIC_MAX_N * GENERIC_MAX_N], li[GENERIC_MAX_N * GENERIC_MAX_N];   // current lower factor
    double rd[GENERIC_MAX_N];                                                        // 1 / diag of the factor
    double v[GENERIC_MAX_N];
};

SYMPA_HD int gix(int n, int i, int j) { return i * n + j; }

// element (i, j) of a symmetric [n, n] plane stored row-major, reading the upper triangle only
SYMPA_HD double sym_at(const double* __restrict__ p, int n, int i, int j) { return i <= j ? p[i * n + j] : p[j * n + i]; }

// Lower Cholesky factor of the Hermitian matrix currently in (hr, hi) (lower triangle read) -> (lr, li, rd).
SYMPA_HD bool generic_chol(GenericWork& w, int n) {
    bool ok = true;
    for (int j = 0; j < n; ++j) {
        double s = w.hr[gix(n, j, j)];
        for (int k = 0; k < j; ++k) s -= w.lr[gix(n, j, k)] * w.lr[gix(n, j, k)] + w.li[gix(n, j, k)] * w.li[gix(n, j, k)];
        ok = ok && (s > 0.0);
        const double r = d_rsqrt(s);
        w.rd[j] = r;
        for (int i = j + 1; i < n; ++i) {
            double tr = w.hr[gix(n, i, j)], ti = w.hi[gix(n, i, j)];
            for (int k = 0; k < j; ++k) {   // minus l_ik conj(l_jk)
                tr -= w.lr[gix(n, i, k)] * w.lr[gix(n, j, k)] + w.li[gix(n, i, k)] * w.li[gix(n, j, k)];
                ti -= w.li[gix(n, i, k)] * w.lr[gix(n, j, k)] - w.lr[gix(n, i, k)] * w.li[gix(n, j, k)];
            }
            w.lr[gix(n, i, j)] = tr * r;
            w.li[gix(n, i, j)] = ti * r;
        }
    }
    return ok;
}

// Loads the matrix whose Cholesky factor the model needs for one endpoint into (hr, hi) lower triangle.
SYMPA_HD void generic_factor_input(GenericWork& w, int n, int model, const double* __restrict__ p) {
    const double* re = p;
    const double* im = p + n * n;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            if (model == MODEL_UPPER) {              // Y = Im z
                w.hr[gix(n, i, j)] = sym_at(im, n, i, j);
                w.hi[gix(n, i, j)] = 0.0;
            } else {                                  // A = I - W W^H,  a_ij = delta_ij - sum_l w_il conj(w_jl)
                double tr = (i == j) ? 1.0 : 0.0, ti = 0.0;
                for (int l = 0; l < n; ++l) {
                    const double ar = sym_at(re, n, i, l), ai = sym_at(im, n, i, l);
                    const double br = sym_at(re, n, j, l), bi = sym_at(im, n, j, l);
                    tr -= ar * br + ai * bi;
                    ti -= ai * br - ar * bi;
                }
                w.hr[gix(n, i, j)] = tr;
                w.hi[gix(n, i, j)] = (i == j) ? 0.0 : ti;
            }
        }
}

SYMPA_HD double pair_distance_generic(GenericWork& w, const double* __restrict__ p1, const double* __restrict__ p2,
                                      int n, int model, int metric, const double* __restrict__ mw, double inv_eps,
                                      double* __restrict__ vvd, int& status) {
    // E = D = Z2 - Z1
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            w.er[gix(n, i, j)] = sym_at(p2, n, i, j) - sym_at(p1, n, i, j);
            w.ei[gix(n, i, j)] = sym_at(p2 + n * n, n, i, j) - sym_at(p1 + n * n, n, i, j);
        }
    // E <- L1^-1 E
    generic_factor_input(w, n, model, p1);
    bool ok = generic_chol(w, n);
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < n; ++i) {
            double tr = w.er[gix(n, i, c)], ti = w.ei[gix(n, i, c)];
            for (int k = 0; k < i; ++k) {
                const double ar = w.lr[gix(n, i, k)], ai = w.li[gix(n, i, k)];
                tr -= ar * w.er[gix(n, k, c)] - ai * w.ei[gix(n, k, c)];
                ti -= ar * w.ei[gix(n, k, c)] + ai * w.er[gix(n, k, c)];
            }
            w.er[gix(n, i, c)] = tr * w.rd[i];
            w.ei[gix(n, i, c)] = ti * w.rd[i];
        }
    // E <- E L2^-T
    generic_factor_input(w, n, model, p2);
    ok = generic_chol(w, n) && ok;
    for (int r = 0; r < n; ++r)
        for (int j = 0; j < n; ++j) {
            double tr = w.er[gix(n, r, j)], ti = w.ei[gix(n, r, j)];
            for (int k = 0; k < j; ++k) {
                const double ar = w.lr[gix(n, j, k)], ai = w.li[gix(n, j, k)];
                tr -= w.er[gix(n, r, k)] * ar - w.ei[gix(n, r, k)] * ai;
                ti -= w.er[gix(n, r, k)] * ai + w.ei[gix(n, r, k)] * ar;
            }
            w.er[gix(n, r, j)] = tr * w.rd[j];
            w.ei[gix(n, r, j)] = ti * w.rd[j];
        }
    // H = E^H E  (full Hermitian storage, both triangles)
    for (int j = 0; j < n; ++j)
        for (int k = j; k < n; ++k) {
            double tr = 0.0, ti = 0.0;
            for (int i = 0; i < n; ++i) {
                tr += w.er[gix(n, i, j)] * w.er[gix(n, i, k)] + w.ei[gix(n, i, j)] * w.ei[gix(n, i, k)];
                ti += w.er[gix(n, i, j)] * w.ei[gix(n, i, k)] - w.ei[gix(n, i, j)] * w.er[gix(n, i, k)];
            }
            w.hr[gix(n, j, k)] = tr; w.hi[gix(n, j, k)] = (j == k) ? 0.0 : ti;
            w.hr[gix(n, k, j)] = tr; w.hi[gix(n, k, j)] = (j == k) ? 0.0 : -ti;
        }
    // cyclic Jacobi on the full Hermitian matrix until ||off|| <= 1e-11 ||diag||
    bool conv = (n == 1);
    for (int sweep = 0; sweep < 30 && n > 1; ++sweep) {
        double off2 = 0.0, diag2 = 0.0;
        for (int j = 0; j < n; ++j) {
            diag2 += w.hr[gix(n, j, j)] * w.hr[gix(n, j, j)];
            for (int k = j + 1; k < n; ++k) off2 += w.hr[gix(n, j, k)] * w.hr[gix(n, j, k)] + w.hi[gix(n, j, k)] * w.hi[gix(n, j, k)];
        }
        conv = !(off2 > 1e-22 * diag2);
        if (wave_all(conv)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double br = w.hr[gix(n, p, q)], bi = w.hi[gix(n, p, q)];
                const double a2 = br * br + bi * bi;
                const double delta = w.hr[gix(n, q, q)] - w.hr[gix(n, p, p)];
                const double ad = fabs(delta) + 1e-150;
                const double qr = d_rsqrt(ad * ad + 4.0 * a2);
                const double c2 = 0.5 * ad * qr + 0.5;
                const double ic = d_rsqrt(c2);
                const double c = c2 * ic;
                const double cu = copysign(qr, delta) * ic;
                const double ua2 = (cu * ic) * a2;
                const double wr = cu * br, wi = cu * bi;
                w.hr[gix(n, p, p)] -= ua2;
                w.hr[gix(n, q, q)] += ua2;
                w.hr[gix(n, p, q)] = 0.0; w.hi[gix(n, p, q)] = 0.0;
                w.hr[gix(n, q, p)] = 0.0; w.hi[gix(n, q, p)] = 0.0;
                for (int k = 0; k < n; ++k) {
                    if (k == p || k == q) continue;
                    const double xr = w.hr[gix(n, k, p)], xi = w.hi[gix(n, k, p)];
                    const double yr = w.hr[gix(n, k, q)], yi = w.hi[gix(n, k, q)];
                    const double nxr = c * xr - wr * yr - wi * yi, nxi = c * xi - wr * yi + wi * yr;
                    const double nyr = c * yr + wr * xr - wi * xi, nyi = c * yi + wr * xi + wi * xr;
                    w.hr[gix(n, k, p)] = nxr; w.hi[gix(n, k, p)] = nxi;
                    w.hr[gix(n, p, k)] = nxr; w.hi[gix(n, p, k)] = -nxi;
                    w.hr[gix(n, k, q)] = nyr; w.hi[gix(n, k, q)] = nyi;
                    w.hr[gix(n, q, k)] = nyr; w.hi[gix(n, q, k)] = -nyi;
                }
            }
    }
    const double scale = (model == MODEL_UPPER) ? 0.25 : 1.0;
    bool finite = true;          // tested before the clamp: fmax would turn a NaN eigenvalue into distance 0
    for (int i = 0; i < n; ++i) {
        finite = finite && d_finite(w.hr[gix(n, i, i)]);
        w.v[i] = vvd_from_sinh2(fmax(w.hr[gix(n, i, i)], 0.0) * scale, inv_eps);
    }
    for (int i = 1; i < n; ++i) {            // insertion sort, ascending
        const double x = w.v[i];
        int j = i - 1;
        while (j >= 0 && w.v[j] > x) { w.v[j + 1] = w.v[j]; --j; }
        w.v[j + 1] = x;
    }
    double out = 0.0;
    if (metric == METRIC_RIEM) { for (int i = 0; i < n; ++i) out += w.v[i] * w.v[i]; out = d_sqrt(out); }
    else if (metric == METRIC_FONE) { for (int i = 0; i < n; ++i) out += w.v[i]; }
    else if (metric == METRIC_FINF) { out = w.v[n - 1]; }
    else if (metric == METRIC_FMIN) { for (int i = 0; i < n; ++i) out += 2.0 * i * w.v[i]; }
    else { for (int i = 0; i < n; ++i) out += fmax(mw[i], 0.0) * w.v[i]; }
    if (!finite) {
        out = __builtin_nan("");
        for (int i = 0; i < n; ++i) w.v[i] = out;
    }
    if (vvd != nullptr) for (int i = 0; i < n; ++i) vvd[i] = w.v[i];
    if (!ok) status |= ST_NOT_PD;
    if (!conv) status |= ST_NO_CONVERGENCE;
    if (!(out == out) || !(fabs(out) <= 1.79e308)) status |= ST_NONFINITE;
    return out;
}

}  // namespace sympa
