// Eigenvectors of a symmetric tridiagonal matrix from its (already computed) eigenvalues by inverse iteration, ONE MATRIX
// PER LANE, everything in statically indexed registers, branch-free inside a vector, a fixed number of iterations -- the
// wave runs it in lockstep.  Restated from the published algorithm of LAPACK's dstein / dlagtf / dlagts (LU factorisation
// of T - lambda I with partial pivoting, solves with perturbed tiny pivots, modified Gram-Schmidt inside clusters of close
// eigenvalues, distinct shifts inside a cluster); no reference counterpart (the reference calls torch.symeig).
//
// Why: the SPD backward needs V of L^-1 (Y - X) L^-T.  QL WITH accumulated rotations cannot run one pair per lane (Z is 256
// doubles at n = 16), so the sixteen-lanes-per-pair kernel runs the scalar QL recurrence redundantly in the sixteen lanes of
// a pair: four (eight) pairs per instruction stream where the forward's eigenvalue-only QL serves 64 -- 80 % of the backward.
// Here the eigenvalues come from the forward's lockstep QL (64 pairs per stream) and an eigenvector costs O(n) per
// eigenvalue, again 64 pairs per stream; each finished vector (16 doubles) leaves through `store`.
//
// Clusters: eigenvalues closer than 1e-3 ||T||_1 form a block (dstein's criterion); a vector is orthogonalised against the
// previous vectors of its block.  The last KEEP vectors are held in registers; a block of more than KEEP + 1 eigenvalues
// raises `big_cluster` and the caller routes that pair to the QL-with-vectors kernel (a generic data set never does; exact
// multiples y = c x do: every eigenvalue equal).
#pragma once

#include "siegel_math.hpp"

namespace sympa {

constexpr int INVIT_KEEP = 3;        // previous vectors of the current block kept in registers
constexpr int INVIT_ITERS = 2;       // solves per eigenvector (dstein: convergence + 2 extra; accurate shifts converge in one)

// deterministic start vectors in (-1, 1) (dstein draws them at random): element j of the start vector of eigenvalue i
constexpr double invit_start(int i, int j) {
    unsigned long long h = 0x9E3779B97F4A7C15ull * (unsigned long long)(i * 131 + j * 7 + 1);
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return ((double)(h >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
}

template <int I, int E, class F>
SYMPA_HD void invit_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        invit_for<I + 1, E>(f);
    }
}

// d[0..M), e[0..M-1) (e[i] = T[i+1][i]; e[M-1] ignored), lam[0..M) ASCENDING.  store(IC, z): IC = integral_constant index of
// the eigenvalue, z = its unit eigenvector.  Returns false when a block of more than INVIT_KEEP + 1 eigenvalues was met.
template <int M, class Store>
SYMPA_HD bool tridiag_eigvecs_invit(const double (&d)[M], const double (&e)[M], const double (&lam)[M], Store&& store) {
    constexpr double EPS = 1.1102230246251565e-16;
    // ||T||_1
    double onenrm = 0.0;
SYMPA_UNROLL
    for (int i = 0; i < M; ++i) {
        const double lo = (i > 0) ? fabs(e[i - 1]) : 0.0;
        const double hi = (i < M - 1) ? fabs(e[i]) : 0.0;
        onenrm = fmax(onenrm, fabs(d[i]) + lo + hi);
    }
    onenrm = fmax(onenrm, 1e-250);           // T = 0 (y = x): eps ||T|| must stay a normal number, its reciprocal finite
    const double ortol = 1e-3 * onenrm;
    const double pivtiny = EPS * onenrm;
    double prev[INVIT_KEEP][M];
SYMPA_UNROLL
    for (int k = 0; k < INVIT_KEEP; ++k)
SYMPA_UNROLL
        for (int j = 0; j < M; ++j) prev[k][j] = 0.0;
    int in_block = 0;          // previous vectors that belong to the current block
    bool ok = true;
    double prev_shift = 0.0;
    invit_for<0, M>([&](auto IC) {
        constexpr int I = IC;
        double xj = lam[I];
        if constexpr (I > 0) {
            const bool close = (lam[I] - lam[I - 1]) < ortol;
            in_block = close ? in_block + 1 : 0;
            // distinct shifts inside a block (dstein): x_j >= x_{j-1} + 10 eps |x_j|
            const double pertol = 10.0 * EPS * fmax(fabs(xj), pivtiny);
            xj = (close && (xj - prev_shift) < pertol) ? prev_shift + pertol : xj;
        }
        prev_shift = xj;
        ok = ok && (in_block <= INVIT_KEEP);
        // ---- LU of T - xj I with partial pivoting (dlagtf): U = diag a, superdiagonal b, second superdiagonal d2;
        //      multipliers c, interchange flags in the bits of `swaps`
        double a[M], b[M], c[M], d2[M];
SYMPA_UNROLL
        for (int k = 0; k < M; ++k) {
            a[k] = d[k] - xj;
            b[k] = (k < M - 1) ? e[k] : 0.0;
            c[k] = (k < M - 1) ? e[k] : 0.0;
            d2[k] = 0.0;
        }
        unsigned swaps = 0u;
SYMPA_UNROLL
        for (int k = 0; k < M - 1; ++k) {
            const bool sw = fabs(c[k]) > fabs(a[k]);
            // no interchange: mult = c / a (a may be 0 only if c is 0 too: then mult = 0)
            const double piv = sw ? c[k] : a[k];
            const double num = sw ? a[k] : c[k];
            const double safe = (piv != 0.0) ? piv : 1.0;
            const double mult = (piv != 0.0) ? num * d_rcp(safe) : 0.0;
            const double ak1 = a[k + 1];
            const double bk = b[k];
            const double bk1 = (k < M - 2) ? b[k + 1] : 0.0;
            // interchange: row k <- old row k+1 = (c, a[k+1], b[k+1]); row k+1 <- old row k - mult * that
            a[k] = piv;
            a[k + 1] = sw ? d_fma(-mult, ak1, bk) : d_fma(-mult, bk, ak1);
            b[k] = sw ? ak1 : bk;
            if (k < M - 2) {
                d2[k] = sw ? bk1 : 0.0;
                b[k + 1] = sw ? -mult * bk1 : bk1;
            }
            c[k] = mult;
            swaps |= sw ? (1u << k) : 0u;
        }
        // reciprocal pivots, tiny ones perturbed (dlagts, job < 0)
        double ra[M];
SYMPA_UNROLL
        for (int k = 0; k < M; ++k) {
            const double ak = (fabs(a[k]) < pivtiny) ? copysign(pivtiny, a[k]) : a[k];
            ra[k] = d_rcp(ak);
        }
        // ---- inverse iteration
        double x[M];
SYMPA_UNROLL
        for (int j = 0; j < M; ++j) x[j] = invit_start(I, j);
SYMPA_UNROLL
        for (int it = 0; it < INVIT_ITERS; ++it) {
            // scale to unit max-norm: the solve multiplies by up to 1 / (eps ||T||)
            double mx = 0.0;
SYMPA_UNROLL
            for (int j = 0; j < M; ++j) mx = fmax(mx, fabs(x[j]));
            const double sc = d_rcp(fmax(mx, 1e-300));
SYMPA_UNROLL
            for (int j = 0; j < M; ++j) x[j] *= sc;
            // forward: y = L^-1 P x
SYMPA_UNROLL
            for (int k = 0; k < M - 1; ++k) {
                const bool sw = (swaps >> k) & 1u;
                const double top = sw ? x[k + 1] : x[k];
                const double bot = sw ? x[k] : x[k + 1];
                x[k] = top;
                x[k + 1] = d_fma(-c[k], top, bot);
            }
            // backward: U z = y
SYMPA_UNROLL
            for (int k = M - 1; k >= 0; --k) {
                double t = x[k];
                if (k < M - 1) t = d_fma(-b[k], x[k + 1], t);
                if (k < M - 2) t = d_fma(-d2[k], x[k + 2], t);
                x[k] = t * ra[k];
            }
            // modified Gram-Schmidt against the previous vectors of the block (unit vectors)
            if constexpr (I > 0) {
SYMPA_UNROLL
                for (int q = 0; q < INVIT_KEEP; ++q) {
                    if (q < I) {
                        double dot = 0.0;
SYMPA_UNROLL
                        for (int j = 0; j < M; ++j) dot = d_fma(x[j], prev[q][j], dot);
                        dot = (in_block > q) ? dot : 0.0;
SYMPA_UNROLL
                        for (int j = 0; j < M; ++j) x[j] = d_fma(-dot, prev[q][j], x[j]);
                    }
                }
            }
        }
        // unit 2-norm (after a max-norm scaling: the iterate may be ~1e16 long)
        double mx = 0.0;
SYMPA_UNROLL
        for (int j = 0; j < M; ++j) mx = fmax(mx, fabs(x[j]));
        const double sc = d_rcp(fmax(mx, 1e-300));
        double n2 = 0.0;
SYMPA_UNROLL
        for (int j = 0; j < M; ++j) { x[j] *= sc; n2 = d_fma(x[j], x[j], n2); }
        const double rn = d_rsqrt(fmax(n2, 1e-300));
SYMPA_UNROLL
        for (int j = 0; j < M; ++j) x[j] *= rn;
        store(IC, x);
        // shift the kept vectors: prev[0] = newest
SYMPA_UNROLL
        for (int q = INVIT_KEEP - 1; q > 0; --q)
SYMPA_UNROLL
            for (int j = 0; j < M; ++j) prev[q][j] = prev[q - 1][j];
SYMPA_UNROLL
        for (int j = 0; j < M; ++j) prev[0][j] = x[j];
    });
    return ok;
}

}  // namespace sympa
