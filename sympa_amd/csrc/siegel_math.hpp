// Per-pair arithmetic of the Siegel-distance hot path, written once and compiled twice:
//   * by hipcc for gfx950 inside the kernels of siegel_dist.hip (the product), one pair per lane,
//     every matrix in registers (all loops below unroll completely, every index is static);
//   * by g++ for tests/hostsim (CPU, `-m "not gpu"` tests only) so the exact kernel arithmetic is
//     checked against the oracle in the container that has no GPU.  hostsim is never shipped or
//     loaded by the product package.
//
// What it replaces in the reference (fedelopez77/sympa), per pair (z1, z2):
//   SiegelManifold.dist            sympa/manifolds/siegel_manifold.py:41-72
//   sm.matrix_sqrt + .inverse()    sympa/math/csym_math.py:509-520, siegel_manifold.py:53
//   sm.bmm3 sandwich               sympa/math/csym_math.py:91-128
//   cayley_transform / sm.inverse  sympa/math/cayley_transform.py:10-24, csym_math.py:197-249
//   inverse_cayley_transform x2    sympa/math/cayley_transform.py:27-40 (bounded_domain.py:27-39)
//   TakagiFactorization.factorize  sympa/math/takagi_factorization.py:66-75
//   v = log((1+d)/clamp(1-d,eps))  siegel_manifold.py:68-70
//   Metric.compute_metric          sympa/manifolds/metrics.py:42-121
//
// The reference evaluates  d_i = singular values of  W = Cayley(Y1^-1/2 (Z2 - X1) Y1^-1/2)  through
// two symmetric eigendecompositions (n x n and 2n x 2n), three LU inverses and 17 matmuls.
// This file evaluates the same quantity through the identity
//
//        sinh(v_i / 2) = d_i / sqrt(1 - d_i^2) = 1/2 * sigma_i( L1^-1 (Z2 - Z1) L2^-T ),   Y_k = L_k L_k^T
//
// (upper half space) and, for the bounded domain, with A_k = I - W_k W_k^H = C_k C_k^H,
//
//        sinh(v_i / 2) = sigma_i( C1^-1 (W2 - W1) C2^-T )
//
// i.e. two Cholesky factorisations, two triangular solves, one n x n Hermitian Gram matrix and a
// cyclic Jacobi eigenvalue iteration on it: no matrix square root, no complex inverse, no 2n x 2n
// problem, no cancellation in 1 - d.  The reference's clamp of (1 - d) at eps is reproduced exactly.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SYMPA_HD __host__ __device__ __forceinline__
#else
#define SYMPA_HD inline
#endif

// Every loop over matrix indices is fully unrolled so that the matrices live in registers (static indices).  One
// translation unit (siegel_bwd_rolled.hip: backward for dims 9..16) defines SYMPA_UNROLL as `nounroll` before including
// these headers: the same arithmetic then runs as rolled loops over per-lane scratch arrays.
#ifndef SYMPA_UNROLL
#define SYMPA_UNROLL _Pragma("unroll")
#endif

namespace sympa {

enum Model : int { MODEL_UPPER = 0, MODEL_BOUNDED = 1 };
enum Metric : int { METRIC_RIEM = 0, METRIC_FONE = 1, METRIC_FINF = 2, METRIC_FMIN = 3, METRIC_WSUM = 4 };

// status bits accumulated per launch (device counter words, see include/sympa_hip.h)
enum StatusBit : int { ST_NOT_PD = 1, ST_NONFINITE = 2, ST_BAD_INDEX = 4, ST_NO_CONVERGENCE = 8 };

// ---------------------------------------------------------------------------------------------
// fp64 primitives.  Measured on MI355X (tools/microbench/fp64_ubench.hip, profiles/r01_fp64_ubench.txt):
// with one wave per SIMD every fp64 VALU op costs ~9 cycles whether dependent or not, the library's
// IEEE sqrt costs ~94 and IEEE division ~71 cycles, while v_rsq_f64 / v_rcp_f64 cost ~15 and return
// ~24 correct bits (5e-8).  One third-order correction step brings those seeds to full fp64
// accuracy (error ~e^3 = 1e-22 plus rounding) in 4-5 more ops, so every sqrt / division of the hot
// path is expressed through them.  Arguments are positive, finite and normal by construction.
// ---------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
SYMPA_HD bool wave_all(bool p) { return __all(p ? 1 : 0) != 0; }
SYMPA_HD double seed_rsq(double x) { return __builtin_amdgcn_rsq(x); }
SYMPA_HD double seed_rcp(double x) { return __builtin_amdgcn_rcp(x); }
SYMPA_HD double d_frexp_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }   // in [0.5, 1)
SYMPA_HD int d_frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
#else
SYMPA_HD bool wave_all(bool p) { return p; }
// the host build perturbs the seeds to the hardware's measured accuracy so that the CPU tests
// exercise the correction steps instead of starting from an exact value
SYMPA_HD double seed_rsq(double x) { return (1.0 / std::sqrt(x)) * (1.0 + 4.0e-8); }
SYMPA_HD double seed_rcp(double x) { return (1.0 / x) * (1.0 - 4.0e-8); }
SYMPA_HD double d_frexp_mant(double x) { int e; return std::frexp(x, &e); }
SYMPA_HD int d_frexp_exp(double x) { int e; (void)std::frexp(x, &e); return e; }
#endif
SYMPA_HD double d_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// neither NaN nor +-Inf (every comparison with a NaN is false)
SYMPA_HD bool d_finite(double x) { return fabs(x) <= 1.79e308; }

// 1/sqrt(x): seed + one Halley step  y <- y (1 + e/2 + 3 e^2/8),  e = 1 - x y^2.      (6 ops)
SYMPA_HD double d_rsqrt(double x) {
    const double y = seed_rsq(x);
    const double e = d_fma(-(x * y), y, 1.0);
    return d_fma(y, e * d_fma(0.375, e, 0.5), y);
}
// 1/x: seed + one second-order step  y <- y (1 + e + e^2),  e = 1 - x y.              (4 ops)
SYMPA_HD double d_rcp(double x) {
    const double y = seed_rcp(x);
    const double e = d_fma(-x, y, 1.0);
    return d_fma(y, d_fma(e, e, e), y);
}
// sqrt(x) for x >= 0 (exact 0 allowed): x * rsqrt(x + tiny).
constexpr double TINY = 1e-300;
SYMPA_HD double d_sqrt(double x) { return x * d_rsqrt(x + TINY); }

// log(1 + u), u >= 0, relative accuracy ~2 ulp for every magnitude of u.
// u < 0.4: f = u exactly (no rounding of 1 + u); otherwise 1 + u = m 2^k with m in [sqrt(1/2), sqrt(2)),
// f = m - 1.  log(1 + f) = f - s (f - R(s^2)),  s = f / (2 + f),  R = the classic 7-term minimax
// series of log((1+s)/(1-s)) - 2s on that interval (Lg1..Lg7 of the fdlibm/msun e_log.c kernel).
SYMPA_HD double d_log1p(double u) {
    const double x = 1.0 + u;
    double m = d_frexp_mant(x);
    int k = d_frexp_exp(x);
    const bool lowhalf = m < 0.70710678118654752440;
    m = lowhalf ? 2.0 * m : m;
    k = lowhalf ? k - 1 : k;
    const bool small = u < 0.4;
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

// A complex n x n matrix in registers: separate real / imaginary planes.
template <int N>
struct CMat {
    double re[N][N];
    double im[N][N];
};

// Lower-triangular factor with reciprocal diagonal kept separately (the solves multiply by it).
template <int N, bool COMPLEX>
struct Tri {
    double re[N][N];   // strictly-lower part used: re[i][j], j < i
    double im[N][N];   // only when COMPLEX
    double rdiag[N];   // 1 / L[i][i]   (diagonal is real positive in both cases)
};

// ---------------------------------------------------------------------------------------------
// Loads.  A point is [2, n, n] fp64 row-major (reference layout, csym_math.py:1-8).  Only the upper
// triangle (i <= j) is read: points on the manifold are symmetric (reference PRE-condition; the
// reference's own symeig(upper=True) also reads only the upper triangle of Y).
// ---------------------------------------------------------------------------------------------
template <int N>
SYMPA_HD void load_point(const double* __restrict__ p, CMat<N>& z) {
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            z.re[i][j] = p[(i <= j) ? i * N + j : j * N + i];
            z.im[i][j] = p[N * N + ((i <= j) ? i * N + j : j * N + i)];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Cholesky of a real SPD matrix (upper model: Y = L L^T).
// ---------------------------------------------------------------------------------------------
template <int N>
SYMPA_HD bool chol_real(const double (&y)[N][N], Tri<N, false>& l) {
    bool ok = true;
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
        double s = y[j][j];
SYMPA_UNROLL
        for (int k = 0; k < j; ++k) s = d_fma(-l.re[j][k], l.re[j][k], s);
        ok = ok && (s > 0.0);
        const double r = d_rsqrt(s);
        l.rdiag[j] = r;
SYMPA_UNROLL
        for (int i = j + 1; i < N; ++i) {
            double t = y[j][i];
SYMPA_UNROLL
            for (int k = 0; k < j; ++k) t = d_fma(-l.re[i][k], l.re[j][k], t);
            l.re[i][j] = t * r;
        }
    }
    return ok;
}

// Cholesky of the Hermitian matrix A = I - W W^H for complex-symmetric W (bounded model).
// (A)_ij = delta_ij - sum_l w_il conj(w_jl);  A = C C^H.
template <int N>
SYMPA_HD bool chol_id_minus_wwh(const CMat<N>& w, Tri<N, true>& c) {
    bool ok = true;
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
        double s = 1.0;
SYMPA_UNROLL
        for (int l = 0; l < N; ++l) {
            s = d_fma(-w.re[j][l], w.re[j][l], s);
            s = d_fma(-w.im[j][l], w.im[j][l], s);
        }
SYMPA_UNROLL
        for (int k = 0; k < j; ++k) {
            s = d_fma(-c.re[j][k], c.re[j][k], s);
            s = d_fma(-c.im[j][k], c.im[j][k], s);
        }
        ok = ok && (s > 0.0);
        const double r = d_rsqrt(s);
        c.rdiag[j] = r;
SYMPA_UNROLL
        for (int i = j + 1; i < N; ++i) {
            // a_ij = - sum_l w_il conj(w_jl)          (i != j)
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int l = 0; l < N; ++l) {
                tr = d_fma(-w.re[i][l], w.re[j][l], tr);
                tr = d_fma(-w.im[i][l], w.im[j][l], tr);
                ti = d_fma(-w.im[i][l], w.re[j][l], ti);
                ti = d_fma(w.re[i][l], w.im[j][l], ti);
            }
            // minus sum_k c_ik conj(c_jk)
SYMPA_UNROLL
            for (int k = 0; k < j; ++k) {
                tr = d_fma(-c.re[i][k], c.re[j][k], tr);
                tr = d_fma(-c.im[i][k], c.im[j][k], tr);
                ti = d_fma(-c.im[i][k], c.re[j][k], ti);
                ti = d_fma(c.re[i][k], c.im[j][k], ti);
            }
            c.re[i][j] = tr * r;
            c.im[i][j] = ti * r;
        }
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------
// E <- L1^-1 * E   (forward substitution, column by column; E is overwritten)
// ---------------------------------------------------------------------------------------------
template <int N, bool COMPLEX>
SYMPA_HD void solve_left(const Tri<N, COMPLEX>& l, CMat<N>& e) {
SYMPA_UNROLL
    for (int c = 0; c < N; ++c) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) {
            double tr = e.re[i][c], ti = e.im[i][c];
SYMPA_UNROLL
            for (int k = 0; k < i; ++k) {
                tr = d_fma(-l.re[i][k], e.re[k][c], tr);
                ti = d_fma(-l.re[i][k], e.im[k][c], ti);
                if (COMPLEX) {
                    tr = d_fma(l.im[i][k], e.im[k][c], tr);
                    ti = d_fma(-l.im[i][k], e.re[k][c], ti);
                }
            }
            e.re[i][c] = tr * l.rdiag[i];
            e.im[i][c] = ti * l.rdiag[i];
        }
    }
}

// E <- E * L2^-T   (solve X L2^T = E row by row; plain transpose, no conjugation)
template <int N, bool COMPLEX>
SYMPA_HD void solve_right_t(const Tri<N, COMPLEX>& l, CMat<N>& e) {
SYMPA_UNROLL
    for (int r = 0; r < N; ++r) {
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            double tr = e.re[r][j], ti = e.im[r][j];
SYMPA_UNROLL
            for (int k = 0; k < j; ++k) {
                tr = d_fma(-e.re[r][k], l.re[j][k], tr);
                ti = d_fma(-e.im[r][k], l.re[j][k], ti);
                if (COMPLEX) {
                    tr = d_fma(e.im[r][k], l.im[j][k], tr);
                    ti = d_fma(-e.re[r][k], l.im[j][k], ti);
                }
            }
            e.re[r][j] = tr * l.rdiag[j];
            e.im[r][j] = ti * l.rdiag[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Hermitian Gram matrix H = E^H E, stored as real diagonal d[] + strictly-upper complex part.
// ---------------------------------------------------------------------------------------------
template <int N>
struct Herm {
    double d[N];
    double re[N][N];   // re[j][k], j < k
    double im[N][N];
};

template <int N>
SYMPA_HD void gram(const CMat<N>& e, Herm<N>& h) {
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
        double s = 0.0;
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) {
            s = d_fma(e.re[i][j], e.re[i][j], s);
            s = d_fma(e.im[i][j], e.im[i][j], s);
        }
        h.d[j] = s;
SYMPA_UNROLL
        for (int k = j + 1; k < N; ++k) {
            double tr = 0.0, ti = 0.0;   // sum_i conj(e_ij) e_ik
SYMPA_UNROLL
            for (int i = 0; i < N; ++i) {
                tr = d_fma(e.re[i][j], e.re[i][k], tr);
                tr = d_fma(e.im[i][j], e.im[i][k], tr);
                ti = d_fma(e.re[i][j], e.im[i][k], ti);
                ti = d_fma(-e.im[i][j], e.re[i][k], ti);
            }
            h.re[j][k] = tr;
            h.im[j][k] = ti;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Cyclic (row-by-row) Jacobi eigenvalue iteration on a Hermitian matrix, eigenvalues only.
// Pivot (p,q), beta = h_pq = a e^{i phi}:  J = [[c, s e^{i phi}], [-s e^{-i phi}, c]],  H <- J^H H J.
// With delta = h_qq - h_pp, r = sqrt(delta^2 + 4 a^2):  cos 2theta = |delta| / r,  sin 2theta = 2 a / r, so
//   c^2 = (1 + |delta|/r) / 2,   s / a = sgn(delta) / (r c),   t a = a^2 / (r c^2)
// which needs two reciprocal square roots, no division and no sqrt(a^2), and has no cancellation.
// The rounds {(0,1),(2,3)}, {(0,2),(1,3)}, ... of the round-robin order touch disjoint index pairs,
// so their parameter chains are independent instruction streams.
// ---------------------------------------------------------------------------------------------
template <int N>
SYMPA_HD void jacobi_rotate(Herm<N>& h, const int p, const int q) {   // p < q, static after unrolling
        {
        const double br = h.re[p][q], bi = h.im[p][q];
        const double a2 = d_fma(br, br, bi * bi);
        const double delta = h.d[q] - h.d[p];
        // |delta| + 1e-150 keeps r2 > 0 (and gives c = 1, w = 0) when beta = delta = 0
        const double ad = fabs(delta) + 1e-150;
        const double qr = d_rsqrt(d_fma(ad, ad, 4.0 * a2));      // 1 / r,  r = sqrt(delta^2 + 4 a^2)
        const double c2 = d_fma(0.5 * ad, qr, 0.5);                // cos^2 = (1 + |delta|/r) / 2
        const double ic = d_rsqrt(c2);
        const double c = c2 * ic;
        const double cu = copysign(qr, delta) * ic;                // s / a, signed
        const double ua2 = (cu * ic) * a2;                         // t * |beta|
        const double wr = cu * br, wi = cu * bi;                   // w = s e^{i phi}
        h.d[p] -= ua2;
        h.d[q] += ua2;
        h.re[p][q] = 0.0;
        h.im[p][q] = 0.0;
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) {
            if (k == p || k == q) continue;
            // x = h_kp, y = h_kq  (conjugate when the stored element is the transposed one)
            double xr, xi, yr, yi;
            if (k < p) { xr = h.re[k][p]; xi = h.im[k][p]; } else { xr = h.re[p][k]; xi = -h.im[p][k]; }
            if (k < q) { yr = h.re[k][q]; yi = h.im[k][q]; } else { yr = h.re[q][k]; yi = -h.im[q][k]; }
            // x' = c x - conj(w) y ;  y' = w x + c y
            const double nxr = d_fma(-wi, yi, d_fma(-wr, yr, c * xr));
            const double nxi = d_fma(wi, yr, d_fma(-wr, yi, c * xi));
            const double nyr = d_fma(-wi, xi, d_fma(wr, xr, c * yr));
            const double nyi = d_fma(wi, xr, d_fma(wr, xi, c * yi));
            if (k < p) { h.re[k][p] = nxr; h.im[k][p] = nxi; } else { h.re[p][k] = nxr; h.im[p][k] = -nxi; }
            if (k < q) { h.re[k][q] = nyr; h.im[k][q] = nyi; } else { h.re[q][k] = nyr; h.im[q][k] = -nyi; }
        }
        }
}

template <int N>
SYMPA_HD void jacobi_sweep(Herm<N>& h) {
    if (N == 4) {   // round-robin: 3 rounds of 2 disjoint rotations
        jacobi_rotate<N>(h, 0, 1); jacobi_rotate<N>(h, 2, 3);
        jacobi_rotate<N>(h, 0, 2); jacobi_rotate<N>(h, 1, 3);
        jacobi_rotate<N>(h, 0, 3); jacobi_rotate<N>(h, 1, 2);
    } else {        // cyclic by rows
SYMPA_UNROLL
        for (int p = 0; p < N - 1; ++p) {
SYMPA_UNROLL
            for (int q = p + 1; q < N; ++q) jacobi_rotate<N>(h, p, q);
        }
    }
}

template <int N>
SYMPA_HD void herm_norms(const Herm<N>& h, double& off2, double& diag2) {
    off2 = 0.0;
    diag2 = 0.0;
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
        diag2 = d_fma(h.d[j], h.d[j], diag2);
SYMPA_UNROLL
        for (int k = j + 1; k < N; ++k) {
            off2 = d_fma(h.re[j][k], h.re[j][k], off2);
            off2 = d_fma(h.im[j][k], h.im[j][k], off2);
        }
    }
}

// Finishing sweep.  Once every pivot's rotation angle is small the second-order shifts
//   d_p -= t|beta|,  d_q += t|beta|     (exact eigenvalues of each 2 x 2 pivot problem)
// are applied to the diagonal WITHOUT rotating the remaining off-diagonal entries (40 % of the cost of a
// full sweep).  What this drops, in Rayleigh-Schroedinger terms:
//   * third order  sum_{j,k} f_ij f_jk f_ki / (D_ij D_ik): needs a triangle of non-zero off-diagonal
//     entries.  For n = 4 the round-robin sweep ends with the rounds {(0,2),(1,3)}, {(0,3),(1,2)}; the last
//     round leaves h_03 = h_12 = 0 exactly and every triangle of four indices contains one of those two
//     edges, so the third-order term VANISHES identically;
//   * fourth order  ~ n^2 t^4 ||H||  with t = max |h_pq / (h_qq - h_pp)|.
// Certificate (jacobi_can_finish): t^2 <= 1e-3 on every pivot (or the pivot is negligible against the
// whole matrix) and off-diagonal mass <= 1e-4 of the diagonal mass  =>  dropped terms <= 16 * 1e-6 ||H||,
// i.e. a certified relative error of the distance below 2e-5 (north_star tolerance: 1e-4).  MEASURED on
// 4 x 65 536 pairs of each benchmark table (init, trained 0.3, trained 1.0) and on every golden vector:
// <= 3e-12, typically 1e-15, because after three sweeps t is ~1e-5 on all but a handful of pivots
// (tools/jacobi_convergence.py reproduces these statistics).  For n > 4 the third-order term does not vanish
// and graded spectra (lambda_max / lambda_min ~ 1e10 near the clamp) need the small eigenvalues to RELATIVE
// accuracy for fone / fmin: a looser pair (t^2 <= 1e-4, off <= 1e-8) measured 8e-8 on the bounded n = 8 golden
// vectors, so the certificate stays at t^2 <= 1e-8, off <= 1e-12 (dropped terms <= 6e-13 ||H||).
template <int N>
SYMPA_HD void jacobi_diag_update(Herm<N>& h, const int p, const int q) {
    const double br = h.re[p][q], bi = h.im[p][q];
    const double a2 = d_fma(br, br, bi * bi);
    const double delta = h.d[q] - h.d[p];
    const double ad = fabs(delta) + 1e-150;
    const double qr = d_rsqrt(d_fma(ad, ad, 4.0 * a2));
    const double c2 = d_fma(0.5 * ad, qr, 0.5);
    // t |beta| = a^2 / (r c^2);  1 / c^2 = 2 - c^2 + O(t^4)
    const double ua2 = (copysign(qr, delta) * a2) * (2.0 - c2);
    h.d[p] -= ua2;
    h.d[q] += ua2;
}

template <int N>
SYMPA_HD void jacobi_final_sweep(Herm<N>& h) {
SYMPA_UNROLL
    for (int p = 0; p < N - 1; ++p) {
SYMPA_UNROLL
        for (int q = p + 1; q < N; ++q) jacobi_diag_update<N>(h, p, q);
    }
}

// Certificate for the finishing sweep (thresholds above): every pivot has t^2 = |h_pq|^2 / delta^2 small, or is
// already negligible against the whole matrix (|h_pq|^2 <= 1e-24 ||diag||^2: exactly or numerically
// degenerate pairs), and the off-diagonal mass is small against the diagonal mass.
template <int N> constexpr double finish_t2() { return N <= 4 ? 1e-3 : 1e-8; }
template <int N> constexpr double finish_off2() { return N <= 4 ? 1e-4 : 1e-12; }
constexpr double FINISH_NEGLIGIBLE2 = 1e-24;

template <int N>
SYMPA_HD bool jacobi_can_finish(const Herm<N>& h) {
    double diag2 = 0.0;
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) diag2 = d_fma(h.d[j], h.d[j], diag2);
    const double floor2 = FINISH_NEGLIGIBLE2 * diag2;
    double off2 = 0.0;
    bool ok = true;
SYMPA_UNROLL
    for (int p = 0; p < N - 1; ++p) {
SYMPA_UNROLL
        for (int q = p + 1; q < N; ++q) {
            const double a2 = d_fma(h.re[p][q], h.re[p][q], h.im[p][q] * h.im[p][q]);
            const double delta = h.d[q] - h.d[p];
            off2 += a2;
            // written so that a NaN (non-finite input) counts as "finished": no extra sweeps are spent on it, the
            // eigenvalues are tested for finiteness afterwards (pair_distance_mats)
            ok = ok && !(a2 > fmax(finish_t2<N>() * delta * delta, floor2));
        }
    }
    return ok && !(off2 > finish_off2<N>() * diag2);
}

constexpr int JACOBI_MAX_SWEEPS = 16;
// sweeps run before the first test: no pair of the measured inputs passes the certificate earlier
template <int N>
constexpr int jacobi_blind_sweeps() { return N <= 2 ? 1 : (N <= 4 ? 3 : 4); }

// Returns false when the sweep cap was hit before the certificate held.
template <int N>
SYMPA_HD bool herm_eigenvalues_jacobi(Herm<N>& h) {
    if (N == 1) return true;
SYMPA_UNROLL
    for (int sweep = 0; sweep < jacobi_blind_sweeps<N>(); ++sweep) jacobi_sweep<N>(h);
    if (N == 2) return true;    // a single rotation diagonalises a 2 x 2 matrix exactly
    bool conv = false;
    for (int sweep = jacobi_blind_sweeps<N>(); sweep < JACOBI_MAX_SWEEPS; ++sweep) {
        conv = jacobi_can_finish<N>(h);
        if (wave_all(conv)) break;
        jacobi_sweep<N>(h);
    }
    jacobi_final_sweep<N>(h);
    return conv;
}

// ---------------------------------------------------------------------------------------------
// n >= 5: Householder tridiagonalisation + square-root-free QL (Pal-Walker-Kahan, the algorithm of LAPACK's
// dsterf) instead of Jacobi sweeps.  A Jacobi sweep costs ~3 n^3 * 6 instructions and n = 8 needs 6-7 of them
// (17 k instructions, 80 % of the kernel); the reduction costs ~8/3 n^3 once and the QL iteration O(n^2) in
// total.  Only |b_k|^2 of the tridiagonal's off-diagonal enters the eigenvalues, so no phases are fixed.
// Everything stays in registers: loops are static, the per-lane active block [l, m] of the QL iteration is
// handled by predication, the iteration ends when every lane of the wave has deflated all off-diagonals.
// ---------------------------------------------------------------------------------------------
template <int N>
SYMPA_HD void herm_get(const Herm<N>& h, int i, int j, double& re, double& im) {   // element (i, j), i != j
    if (i < j) { re = h.re[i][j]; im = h.im[i][j]; } else { re = h.re[j][i]; im = -h.im[j][i]; }
}

template <int N>
SYMPA_HD void herm_tridiagonalize(Herm<N>& h, double (&a)[N], double (&b2)[N]) {
SYMPA_UNROLL
    for (int k = 0; k < N - 2; ++k) {
        constexpr int dummy = 0; (void)dummy;
        // x_i = H[i][k], i = k+1 .. N-1
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
        b2[k] = n2;
        const double nx = d_sqrt(n2);                    // ||x||
        const double ix0 = d_rsqrt(x02 + TINY);
        const double ax0 = x02 * ix0;                    // |x0|
        const bool x0zero = !(x02 > 0.0);
        const double pr = x0zero ? 1.0 : vr[k + 1] * ix0, pi = x0zero ? 0.0 : vi[k + 1] * ix0;   // phase of x0
        // v = x + phase ||x|| e1 ;  beta = 2 / ||v||^2 = 1 / (||x|| (||x|| + |x0|)); no reflection when sig2 = 0
        vr[k + 1] = pr * (ax0 + nx);
        vi[k + 1] = pi * (ax0 + nx);
        const double beta = (sig2 > 0.0) ? d_rcp(nx * (nx + ax0)) : 0.0;
        // p = beta A v
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
            qr[i] = beta * tr; qi[i] = beta * ti;
            kk = d_fma(vr[i], qr[i], kk); kk = d_fma(vi[i], qi[i], kk);     // Re(v^H p)
        }
        kk *= 0.5 * beta;
SYMPA_UNROLL
        for (int i = k + 1; i < N; ++i) { qr[i] = d_fma(-kk, vr[i], qr[i]); qi[i] = d_fma(-kk, vi[i], qi[i]); }
        // A <- A - v q^H - q v^H
SYMPA_UNROLL
        for (int i = k + 1; i < N; ++i) {
            h.d[i] -= 2.0 * d_fma(vr[i], qr[i], vi[i] * qi[i]);
SYMPA_UNROLL
            for (int j = i + 1; j < N; ++j) {
                // v_i conj(q_j) + q_i conj(v_j)
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
    b2[N - 2] = d_fma(h.re[N - 2][N - 1], h.re[N - 2][N - 1], h.im[N - 2][N - 1] * h.im[N - 2][N - 1]);
    b2[N - 1] = 0.0;
}

// Eigenvalues of the real symmetric tridiagonal (d, e2 = squared off-diagonals), overwriting d.
template <int N>
SYMPA_HD bool tridiag_ql(double (&d)[N], double (&e2)[N]) {
    constexpr double TOL = 1.3e-32;     // eps^2: dsterf's |e|^2 <= eps^2 |d_i d_{i+1}|
    int l = 0;
    bool done = false;
    for (int iter = 0; iter < 40 * N; ++iter) {
SYMPA_UNROLL
        for (int i = 0; i < N - 1; ++i)
            if (e2[i] <= TOL * fabs(d[i] * d[i + 1]) + 1e-290) e2[i] = 0.0;
SYMPA_UNROLL
        for (int i = 0; i < N - 1; ++i)
            if (l == i && e2[i] == 0.0) l = i + 1;
        done = l >= N - 1;
        if (wave_all(done)) break;
        int m = N - 1;
SYMPA_UNROLL
        for (int i = N - 2; i >= 0; --i)
            if (i >= l && e2[i] == 0.0) m = i;
        double dl = 0.0, dl1 = 0.0, el = 1.0, dm = 0.0;
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) {
            if (i == l) dl = d[i];
            if (i == l + 1) dl1 = d[i];
            if (i == m) dm = d[i];
            if (i < N - 1 && i == l) el = e2[i];
        }
        if (done) el = 1.0;
        // Wilkinson shift from the leading 2 x 2 of the block
        const double irte = d_rsqrt(el);
        const double rte = el * irte;
        double sg = 0.5 * (dl1 - dl) * irte;
        const double rr = d_sqrt(d_fma(sg, sg, 1.0));
        const double sigma = dl - rte * d_rcp(sg + copysign(rr, sg));
        double c = 1.0, sn = 0.0, gamma = dm - sigma, p = gamma * gamma;
SYMPA_UNROLL
        for (int i = N - 2; i >= 0; --i) {
            if (!done && i >= l && i <= m - 1) {
                const double bb = e2[i];
                const double r = p + bb;
                if (i < N - 2 && i != m - 1) e2[i + 1] = sn * r;
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
        }
        if (!done) {
SYMPA_UNROLL
            for (int i = 0; i < N; ++i) {
                if (i == l) d[i] = sigma + gamma;
                if (i < N - 1 && i == l) e2[i] = sn * p;
            }
        }
    }
    return done;
}

// The same iteration on a fixed schedule: deflate position l = 0, 1, ... in turn, every lane of the wave sweeping
// the block [l, N-1] until the slowest lane has e_l^2 negligible.  A lane that is already there keeps sweeping with
// the shift d_l: an orthogonal similarity of its remaining block [l+1, N-1] that leaves d_l alone, i.e. harmless --
// so the sweep needs NO per-lane predication (the [l, m] bookkeeping of tridiag_ql costs more than its arithmetic:
// ~1000 instructions per iteration at N = 16, ~50 iterations for the slowest of 64 lanes), its length shrinks with
// l, and the instruction count drops ~5x.  Interior off-diagonals that become negligible are simply rotated
// through (c = 1, s = 0); p + b = 0 is the only new case and means "no rotation".
// Lanes that are through with position L do not idle until the slowest one is: they take their shift from the
// first position in {L+1, L+2} that still has a non-negligible off-diagonal (a deflated e^2 = 0 makes the
// sweep pass through that position with c = 1, s = 0: an identity), so they arrive at the next stages converged.
// (a NaN off-diagonal counts as negligible: the iteration does not spin on non-finite input, which is caught by the
// finiteness test of the eigenvalues afterwards)
// Deflation threshold on e^2 (round 6).  dsterf's eps^2 |d_l d_l+1| asks the off-diagonal itself down to eps; what an eigenvalue
// sees of a neglected off-diagonal is SECOND order, e^2 / gap.  With 1e-20 |d_l d_l+1| that is 1e-20 d / gap relative (1e-12 at a
// relative gap of 1e-8; |e| = 1e-10 d bounds it in a cluster, where the metrics sum the cluster anyway), and the slowest lane of a
// wave often saves the last sweep of a stage: packed forward per 262 144 pairs upper n = 8 124.6 -> 120.9 us, n = 7 98.5 -> 94.7,
// n = 6 77.4 -> 75.5, spd n = 16 per 1 048 576 pairs 1 136 -> 1 097 us (same box, interleaved; 1e-16: no further gain).  Every
// golden, the 50-digit values and the graded-spectrum forward tests are unchanged or closer (profiles/r06_ql_deflation.txt).
// FORWARD only (eigenvalues are all that leaves): the iterations that accumulate or feed eigenVECTORS (backward: tridiag_ql_vectors,
// the inverse-iteration routes) keep eps^2 -- a vector sees a neglected off-diagonal in FIRST order, e / gap.
constexpr double QL_DEFLATE_TOL = 1e-20;
SYMPA_HD bool ql_negligible(double e2, double da, double db) { return !(e2 > 1.3e-32 * fabs(da * db) + 1e-290); }
template <bool VALUES_ONLY>
SYMPA_HD bool ql_deflated(double e2, double da, double db) {
    return !(e2 > (VALUES_ONLY ? QL_DEFLATE_TOL : 1.3e-32) * fabs(da * db) + 1e-290);
}

template <int N, int L, bool VALUES_ONLY>
SYMPA_HD bool tridiag_ql_stage(double (&d)[N], double (&e2)[N]) {
    bool conv = false;
    for (int it = 0; it < 60; ++it) {
        conv = ql_deflated<VALUES_ONLY>(e2[L], d[L], d[L + 1]);
        if (wave_all(conv)) break;
        double dl = d[L], dl1 = d[L + 1], el = e2[L];
        bool idle = conv;
        e2[L] = conv ? 0.0 : e2[L];
        if constexpr (L + 1 <= N - 2) {
            const bool c1 = ql_deflated<VALUES_ONLY>(e2[L + 1], d[L + 1], d[L + 2]);
            dl = idle ? d[L + 1] : dl;
            dl1 = idle ? d[L + 2] : dl1;
            el = idle ? e2[L + 1] : el;
            idle = idle && c1;
            e2[L + 1] = idle ? 0.0 : e2[L + 1];
            if constexpr (L + 2 <= N - 2) {
                const bool c2 = ql_deflated<VALUES_ONLY>(e2[L + 2], d[L + 2], d[L + 3]);
                dl = idle ? d[L + 2] : dl;
                dl1 = idle ? d[L + 3] : dl1;
                el = idle ? e2[L + 2] : el;
                idle = idle && c2;
                e2[L + 2] = idle ? 0.0 : e2[L + 2];
            }
        }
        // Wilkinson shift from the 2 x 2 at the lane's position; an idle lane sweeps with the eigenvalue itself
        el = idle ? 1.0 : el;
        const double irte = d_rsqrt(el);
        const double rte = el * irte;
        const double sg = 0.5 * (dl1 - dl) * irte;
        const double rr = d_sqrt(d_fma(sg, sg, 1.0));
        double sigma = dl - rte * d_rcp(sg + copysign(rr, sg));
        sigma = idle ? dl : sigma;
        double c = 1.0, sn = 0.0, gamma = d[N - 1] - sigma, p = gamma * gamma;
SYMPA_UNROLL
        for (int i = N - 2; i >= L; --i) {
            const double bb = e2[i];
            const double r = p + bb;
            if (i != N - 2) e2[i + 1] = sn * r;
            const double oldc = c;
            // r = 0 (p = b = 0: decoupled and converged) must give c = 1, s = 0: with rs = max(r, tiny),
            // c = (p + (rs - r)) / rs is p / r whenever r > tiny and tiny / tiny = 1 for r = 0
            const double rs = fmax(r, TINY);
            const double ir = d_rcp(rs);
            c = (p + (rs - r)) * ir;
            sn = bb * ir;
            const double oldgam = gamma;
            const double alpha = d[i];
            gamma = d_fma(c, alpha - sigma, -sn * oldgam);
            d[i + 1] = oldgam + (alpha - gamma);
            const double pn = gamma * gamma * d_rcp(c);
            const double pz = oldc * bb;
            p = (c != 0.0) ? pn : pz;
        }
        e2[L] = sn * p;
        d[L] = sigma + gamma;
    }
    return conv;
}

// VALUES_ONLY = true: the forward's eigenvalues (deflation at QL_DEFLATE_TOL); false (default): eps^2, for callers whose
// eigenvalues feed eigenvectors or spectral weights of a backward
template <int N, int L = 0, bool VALUES_ONLY = false>
SYMPA_HD bool tridiag_ql_lockstep(double (&d)[N], double (&e2)[N]) {
    if constexpr (L >= N - 1) {
        return true;
    } else {
        const bool here = tridiag_ql_stage<N, L, VALUES_ONLY>(d, e2);
        const bool rest = tridiag_ql_lockstep<N, L + 1, VALUES_ONLY>(d, e2);
        return here && rest;
    }
}

// Eigenvalues of H into h.d[]: Jacobi for n <= 4, tridiagonal QL for n >= 5.
// (n = 4 through Householder + lockstep QL instead of Jacobi was measured in round 5, profiles/r05_n4_eigen_ab.txt: 9.10 -> 10.19 us
// per 65 536 pairs, the fused 20-step launch 88.0 -> 96.8 us, and the `far` golden fails at 1e-6: Jacobi stays.)
template <int N>
SYMPA_HD bool herm_eigenvalues(Herm<N>& h) {
    if constexpr (N <= 4) {
        return herm_eigenvalues_jacobi<N>(h);
    } else {
        double a[N], b2[N];
        herm_tridiagonalize<N>(h, a, b2);
        const bool ok = tridiag_ql_lockstep<N, 0, true>(a, b2);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) h.d[i] = a[i];
        return ok;
    }
}

// ---------------------------------------------------------------------------------------------
// v = log((1+d)/max(1-d, eps)) from lambda = sinh^2(v/2) = d^2/(1-d^2)   (siegel_manifold.py:68-70)
//   s = sqrt(lambda), c = sqrt(1+lambda):  d = s/c,  (1+d)/(1-d) = (c+s)^2 = 1 + u,
//   u = 2 (lambda + sqrt(lambda (1 + lambda))).  The clamp is active iff  u > (1+d)/eps - 1.
// ---------------------------------------------------------------------------------------------
SYMPA_HD double vvd_from_sinh2(double lambda, double inv_eps) {
    // u = (c+s)^2 - 1 = 2 s^2 + 2 s c = 2 (lambda + sqrt(lambda + lambda^2)): one square root
    double u = 2.0 * (lambda + d_sqrt(d_fma(lambda, lambda, lambda)));
    // (1+d)/eps >= 1/eps, so the clamp can only bite when u + 1 >= 1/eps: rare, wave-divergent branch
    if (u >= inv_eps - 1.0) {
        const double d = d_sqrt(lambda * d_rcp(1.0 + lambda));     // d = s / c
        const double clamped = d_fma(1.0 + d, inv_eps, -1.0);
        u = u < clamped ? u : clamped;
    }
    return d_log1p(u);
}

template <int N>
SYMPA_HD void sort_ascending(double (&v)[N]) {
    // odd-even transposition network: N rounds, static indices
SYMPA_UNROLL
    for (int round = 0; round < N; ++round) {
SYMPA_UNROLL
        for (int i = (round & 1); i + 1 < N; i += 2) {
            const double lo = fmin(v[i], v[i + 1]);
            const double hi = fmax(v[i], v[i + 1]);
            v[i] = lo;
            v[i + 1] = hi;
        }
    }
}

// Metric.compute_metric on the ascending vector v (metrics.py:42-121). `w` = learnable wsum weights.
template <int N>
SYMPA_HD double reduce_metric(double (&v)[N], int metric, const double* __restrict__ w) {
    double acc = 0.0;
    if (metric == METRIC_RIEM) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) acc = d_fma(v[i], v[i], acc);
        return d_sqrt(acc);
    }
    if (metric == METRIC_FONE) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) acc += v[i];
        return acc;
    }
    if (metric == METRIC_FINF) {
        acc = v[0];
SYMPA_UNROLL
        for (int i = 1; i < N; ++i) acc = fmax(acc, v[i]);
        return acc;
    }
    sort_ascending<N>(v);
    if (metric == METRIC_FMIN) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) acc = d_fma(2.0 * i, v[i], acc);
        return acc;
    }
    // METRIC_WSUM: sum relu(w_i) v_i
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) acc = d_fma(fmax(w[i], 0.0), v[i], acc);
    return acc;
}

// ---------------------------------------------------------------------------------------------
// One pair, start to finish.  p1/p2 point at [2,n,n] fp64 points.  Returns the metric value; the
// ascending vector-valued distance is written to vvd (if non-null) and status bits are OR-ed.
// ---------------------------------------------------------------------------------------------
// Second half of a pair: from E (sinh(v_i / 2) = sigma_i(E) / 2 upper, sigma_i(E) bounded) to the metric value.
// From the Hermitian Gram matrix H = E^H E to the metric value (the packed forward evaluates this half in a kernel of its own,
// csrc/siegel_packed_kernel.hpp: H is 2 n^2 registers where E and the factors are 6 n^2, so it runs two waves per SIMD).
template <int N, int MODEL>
SYMPA_HD double distance_from_h(Herm<N>& h, const bool ok, int metric, const double* __restrict__ w,
                                double inv_eps, double* __restrict__ vvd, int& status) {
    const bool conv = herm_eigenvalues<N>(h);

    const double scale = (MODEL == MODEL_UPPER) ? 0.25 : 1.0;
    double v[N];
    // A NaN / Inf anywhere in the two points reaches H through E = L1^-1 (Z2 - Z1) L2^-T and from there the
    // eigenvalues; fmax(lambda, 0) below would turn a NaN into distance 0, so finiteness is tested BEFORE the clamp
    // (the reference yields NaN and fails its assert, siegel_manifold.py:64-66).
    bool finite = true;
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {
        finite = finite && d_finite(h.d[i]);
        v[i] = vvd_from_sinh2(fmax(h.d[i], 0.0) * scale, inv_eps);
    }
    if (!finite) {
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) v[i] = __builtin_nan("");
    }

    if (vvd != nullptr) {
        sort_ascending<N>(v);
SYMPA_UNROLL
        for (int i = 0; i < N; ++i) vvd[i] = v[i];
    }
    double out = reduce_metric<N>(v, metric, w);
    if (!finite) out = __builtin_nan("");       // (the max / min based metrics drop a NaN operand)
    if (!ok) status |= ST_NOT_PD;
    if (!conv) status |= ST_NO_CONVERGENCE;
    if (!d_finite(out)) status |= ST_NONFINITE;
    return out;
}

template <int N, int MODEL>
SYMPA_HD double distance_from_e(const CMat<N>& e, const bool ok, int metric, const double* __restrict__ w,
                                double inv_eps, double* __restrict__ vvd, int& status) {
    Herm<N> h;
    gram<N>(e, h);
    return distance_from_h<N, MODEL>(h, ok, metric, w, inv_eps, vvd, status);
}

template <int N, int MODEL>
SYMPA_HD double pair_distance_mats(const CMat<N>& z1, const CMat<N>& z2, int metric, const double* __restrict__ w,
                                   double inv_eps, double* __restrict__ vvd, int& status) {
    CMat<N> e;
    bool ok;
    {
        if (MODEL == MODEL_UPPER) {
            Tri<N, false> l1, l2;
            ok = chol_real<N>(z1.im, l1);
            ok = chol_real<N>(z2.im, l2) && ok;
SYMPA_UNROLL
            for (int i = 0; i < N; ++i)
SYMPA_UNROLL
                for (int j = 0; j < N; ++j) {
                    e.re[i][j] = z2.re[i][j] - z1.re[i][j];
                    e.im[i][j] = z2.im[i][j] - z1.im[i][j];
                }
            solve_left<N, false>(l1, e);
            solve_right_t<N, false>(l2, e);
        } else {
            Tri<N, true> c1, c2;
            ok = chol_id_minus_wwh<N>(z1, c1);
            ok = chol_id_minus_wwh<N>(z2, c2) && ok;
SYMPA_UNROLL
            for (int i = 0; i < N; ++i)
SYMPA_UNROLL
                for (int j = 0; j < N; ++j) {
                    e.re[i][j] = z2.re[i][j] - z1.re[i][j];
                    e.im[i][j] = z2.im[i][j] - z1.im[i][j];
                }
            solve_left<N, true>(c1, e);
            solve_right_t<N, true>(c2, e);
        }
    }
    return distance_from_e<N, MODEL>(e, ok, metric, w, inv_eps, vvd, status);
}

// ---------------------------------------------------------------------------------------------
// Packed points for the all-pairs matrix (Runner.build_distance_matrix, runner.py:142-154): every point enters
// N pairs, so its factor is computed ONCE -- Y = L L^T (upper) / I - W W^H = C C^H (bounded) -- and stored INVERTED
// next to the point's upper triangle:
//     [ Re z (i <= j, row-major) | Im z (i <= j) | A = L^-1: diagonal, then the strict lower part (re[, im]) ]
// PACK_LEN doubles per point.  A pair then costs E = A1 (Z2 - Z1) A2^T (two triangular products, no division, no
// square root) + distance_from_e.
// ---------------------------------------------------------------------------------------------
template <int N, int MODEL>
struct PointPack {
    static constexpr int TRI = N * (N + 1) / 2;
    static constexpr int LOW = N * (N - 1) / 2;
    static constexpr int OFF_IM = TRI;
    static constexpr int OFF_DIAG = 2 * TRI;
    static constexpr int OFF_LRE = 2 * TRI + N;
    static constexpr int OFF_LIM = 2 * TRI + N + LOW;                       // bounded only
    static constexpr int LEN = 2 * TRI + N + (MODEL == MODEL_UPPER ? LOW : 2 * LOW);
};
SYMPA_HD constexpr int tri_index(int n, int i, int j) { return i * n - i * (i - 1) / 2 + (j - i); }   // i <= j, row-major upper
SYMPA_HD constexpr int low_index(int i, int j) { return i * (i - 1) / 2 + j; }                          // j < i, row-major strict lower

// Packs one point; returns false when its factor does not exist (point outside the manifold).
template <int N, int MODEL>
SYMPA_HD bool pack_point(const CMat<N>& z, double (&p)[PointPack<N, MODEL>::LEN]) {
    using P = PointPack<N, MODEL>;
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = i; j < N; ++j) {
            p[tri_index(N, i, j)] = z.re[i][j];
            p[P::OFF_IM + tri_index(N, i, j)] = z.im[i][j];
        }
    Tri<N, MODEL != MODEL_UPPER> l;
    bool ok;
    if constexpr (MODEL == MODEL_UPPER) ok = chol_real<N>(z.im, l);
    else ok = chol_id_minus_wwh<N>(z, l);
    // A = L^-1, column by column:  A[j][j] = 1 / L[j][j],   A[i][j] = -(1 / L[i][i]) sum_{k=j}^{i-1} L[i][k] A[k][j]
    double ar[N][N], ai[N][N];
SYMPA_UNROLL
    for (int j = 0; j < N; ++j) {
        ar[j][j] = l.rdiag[j];
        ai[j][j] = 0.0;
SYMPA_UNROLL
        for (int i = j + 1; i < N; ++i) {
            double tr = 0.0, ti = 0.0;
SYMPA_UNROLL
            for (int k = j; k < i; ++k) {
                tr = d_fma(l.re[i][k], ar[k][j], tr);
                if (MODEL != MODEL_UPPER) {
                    tr = d_fma(-l.im[i][k], ai[k][j], tr);
                    ti = d_fma(l.re[i][k], ai[k][j], ti);
                    ti = d_fma(l.im[i][k], ar[k][j], ti);
                }
            }
            ar[i][j] = -tr * l.rdiag[i];
            ai[i][j] = -ti * l.rdiag[i];
        }
    }
SYMPA_UNROLL
    for (int i = 0; i < N; ++i) {
        p[P::OFF_DIAG + i] = ar[i][i];
SYMPA_UNROLL
        for (int j = 0; j < i; ++j) {
            p[P::OFF_LRE + low_index(i, j)] = ar[i][j];
            if constexpr (MODEL != MODEL_UPPER) p[P::OFF_LIM + low_index(i, j)] = ai[i][j];
        }
    }
    return ok;
}

// E = A1 (Z2 - Z1) A2^T from two packed points.  P1 / P2 are any indexable sources of doubles (registers, scalar
// registers for the wave-uniform row point of the all-pairs kernel, or a column of an LDS tile for the dims-8 column
// points).  Everything happens IN PLACE in e, one n x n complex matrix: D = Z2 - Z1, then T = A1 D with the rows taken
// from the last to the first (row r of T needs rows k <= r of D only), then E = T A2^T with the columns taken from the last
// to the first -- the products of an 8 x 8 pair then fit the register file next to nothing else (the ascending order kept
// D, T and E alive together: 384 doubles).  Every entry of p1 / p2 is read once (twice for the points themselves).
// DIFF: the point part of p2 already holds Z2 - Z1 (the packed forward subtracts the chunks of the first point as they arrive:
// no second copy of a point's triangles is ever alive) and the point part of p1 is not read.
template <int N, int MODEL, bool DIFF = false, class P1, class P2>
SYMPA_HD void e_from_packed(const P1& p1, const P2& p2, CMat<N>& e) {
    using P = PointPack<N, MODEL>;
    constexpr bool CPLX = MODEL != MODEL_UPPER;
    // D = Z2 - Z1 (symmetric: upper triangle)
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = i; j < N; ++j) {
            const double xr = DIFF ? p2[tri_index(N, i, j)] : p2[tri_index(N, i, j)] - p1[tri_index(N, i, j)];
            const double xi = DIFF ? p2[P::OFF_IM + tri_index(N, i, j)]
                                   : p2[P::OFF_IM + tri_index(N, i, j)] - p1[P::OFF_IM + tri_index(N, i, j)];
            e.re[i][j] = xr; e.im[i][j] = xi;
            e.re[j][i] = xr; e.im[j][i] = xi;
        }
    // T = A1 D   (A1 lower triangular), rows N-1 .. 0
SYMPA_UNROLL
    for (int rr = 0; rr < N; ++rr) {
        const int r = N - 1 - rr;
        const double dg = p1[P::OFF_DIAG + r];
        double lr[N], li[N];
SYMPA_UNROLL
        for (int k = 0; k < r; ++k) {
            lr[k] = p1[P::OFF_LRE + low_index(r, k)];
            if constexpr (CPLX) li[k] = p1[P::OFF_LIM + low_index(r, k)];
        }
SYMPA_UNROLL
        for (int c = 0; c < N; ++c) {
            double xr = dg * e.re[r][c], xi = dg * e.im[r][c];
SYMPA_UNROLL
            for (int k = 0; k < r; ++k) {
                xr = d_fma(lr[k], e.re[k][c], xr);
                xi = d_fma(lr[k], e.im[k][c], xi);
                if constexpr (CPLX) {
                    xr = d_fma(-li[k], e.im[k][c], xr);
                    xi = d_fma(li[k], e.re[k][c], xi);
                }
            }
            e.re[r][c] = xr;
            e.im[r][c] = xi;
        }
    }
    // E = T A2^T (plain transpose):  E[r][c] = sum_{k <= c} T[r][k] A2[c][k], columns N-1 .. 0
SYMPA_UNROLL
    for (int cc = 0; cc < N; ++cc) {
        const int c = N - 1 - cc;
        const double dg = p2[P::OFF_DIAG + c];
        double lr[N], li[N];
SYMPA_UNROLL
        for (int k = 0; k < c; ++k) {
            lr[k] = p2[P::OFF_LRE + low_index(c, k)];
            if constexpr (CPLX) li[k] = p2[P::OFF_LIM + low_index(c, k)];
        }
SYMPA_UNROLL
        for (int r = 0; r < N; ++r) {
            double xr = e.re[r][c] * dg, xi = e.im[r][c] * dg;
SYMPA_UNROLL
            for (int k = 0; k < c; ++k) {
                xr = d_fma(e.re[r][k], lr[k], xr);
                xi = d_fma(e.im[r][k], lr[k], xi);
                if constexpr (CPLX) {
                    xr = d_fma(-e.im[r][k], li[k], xr);
                    xi = d_fma(e.re[r][k], li[k], xi);
                }
            }
            e.re[r][c] = xr;
            e.im[r][c] = xi;
        }
    }
}

// Same, reading the two points straight from memory (lane-per-row loads; used for n > 4 and on the host).
template <int N, int MODEL>
SYMPA_HD double pair_distance(const double* __restrict__ p1, const double* __restrict__ p2, int metric,
                              const double* __restrict__ w, double inv_eps, double* __restrict__ vvd, int& status) {
    CMat<N> z1, z2;
    load_point<N>(p1, z1);
    load_point<N>(p2, z2);
    return pair_distance_mats<N, MODEL>(z1, z2, metric, w, inv_eps, vvd, status);
}

}  // namespace sympa
