// Upper-half / bounded models for 9 <= n <= 16: sixteen lanes per pair (the layout and the DPP machinery of spd_coop.hpp),
// complex, instantiated for every matrix size M = n (no padding: the routines are templates over M, see spd_coop.hpp).
//
// One lane per pair (siegel_math_generic.hpp) keeps E, H and the Cholesky factors of a 16 x 16 pair in scratch memory
// (12 KB per lane) and runs at 3-15 M pairs/s.  Here lane r of a group of 16 owns row r of each REAL matrix of its pair
// -- X1, Y1, X2, Y2, then Re/Im of D = Z2 - Z1, of E = L1^-1 D L2^-T and of H = E^H E; lanes r >= n of a group are
// phantoms (spd_coop.hpp).
//
//   Cholesky Y1 = L1 L1^T, Y2 = L2 L2^T                                  (cholesky_rows, twice)
//   W = D L2^-T, transpose, E^T = W^T L1^-T            for Re and Im     (real factors: the two planes do not mix)
//   H = E^H E    from the columns of E held one per lane:  H[i][j] = sum_k conj(E[k][i]) E[k][j]
//   complex Householder tridiagonalisation of H, reflector taken from column k (one element per lane)
//   (d, |b|^2) kept by lane t of the group in round t;  after 16 rounds: lockstep QL, one pair per lane,
//   v_i = vvd(lambda_i / 4), metric.
//
// Same formulas as pair_distance_mats<N, MODEL_UPPER> (siegel_math.hpp); checked on the GPU against the runtime-n kernel
// (SYMPA_FLAG_GENERIC) and against the oracle.
#pragma once

#include "spd_coop.hpp"

namespace siegel_coop {

using spd_coop::bcast;
using spd_coop::fmac_bc;
using spd_coop::fnmac_bc;
using spd_coop::group_sum;
using spd_coop::settle;
using spd_coop::sfor;
constexpr int N = spd_coop::N;

// H = E^H E.  er/ei: my column of E (element k = E[k][me]).  Row `me` of H comes out in hr/hi.
template <int M>
__device__ __forceinline__ void gram_columns(double (&sr)[M], double (&si)[M], double (&hr)[M], double (&hi)[M]) {
    sfor<0, M>([&](auto K) { sr[K] = settle(sr[K]); si[K] = settle(si[K]); });
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        sfor<0, M>([&](auto K) {
            constexpr int k = K;
            // conj(E[k][i]) E[k][j] = (er_i er_j + ei_i ei_j) + i (er_i ei_j - ei_i er_j)
            fmac_bc<j>(a0, sr[k], sr[k]);
            fmac_bc<j>(a1, si[k], si[k]);
            fmac_bc<j>(b0, si[k], sr[k]);
            fnmac_bc<j>(b1, sr[k], si[k]);
        });
        hr[j] = a0 + a1;
        hi[j] = b0 + b1;
    });
}

// ---- bounded model: A = I - W W^H = C C^H with complex factors -------------------------------------------------
// A = I - W W^H from the rows of the complex-symmetric W held one per lane:  A[i][j] = delta_ij - sum_l W[i][l] conj(W[j][l])
template <int M>
__device__ __forceinline__ void id_minus_wwh_rows(double (&wr)[M], double (&wi)[M], double (&ar)[M], double (&ai)[M],
                                                  const int r) {
    sfor<0, M>([&](auto L) { wr[L] = settle(wr[L]); wi[L] = settle(wi[L]); });
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        double a0 = (r == j) ? 1.0 : 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        sfor<0, M>([&](auto L) {
            constexpr int l = L;
            fnmac_bc<j>(a0, wr[l], wr[l]);      // - wr_i wr_j
            fnmac_bc<j>(a1, wi[l], wi[l]);      // - wi_i wi_j
            fnmac_bc<j>(b0, wr[l], wi[l]);      // - wi_i wr_j
            fmac_bc<j>(b1, wi[l], wr[l]);       // + wr_i wi_j
        });
        ar[j] = a0 + a1;
        ai[j] = b0 + b1;
    });
}

// Cholesky A = C C^H of the Hermitian matrix held one row per lane (complex, right-looking, in place): after step j,
// registers j of lane i >= j hold C[i][j]; the diagonal is real, rd[j] = 1 / C[j][j].
template <int M>
__device__ __forceinline__ bool ccholesky_rows(double (&xr)[M], double (&xi)[M], double (&rd)[M]) {
    bool pd = true;
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        const double piv = bcast<j>(settle(xr[j]));
        pd = pd && (piv > 0.0);
        const double rr = sympa::d_rsqrt(piv);
        rd[j] = rr;
        xr[j] = settle(xr[j] * rr);
        xi[j] = settle(xi[j] * rr);
        sfor<j + 1, M>([&](auto K) {
            constexpr int k = K;
            // X[i][k] -= C[i][j] conj(C[k][j])
            fnmac_bc<k>(xr[k], xr[j], xr[j]);
            fnmac_bc<k>(xr[k], xi[j], xi[j]);
            fnmac_bc<k>(xi[k], xr[j], xi[j]);
            fmac_bc<k>(xi[k], xi[j], xr[j]);
        });
    });
    return pd;
}

// a <- a C^-T (plain transpose) for complex rows held one per lane:  a[j] = (a[j] - sum_{k<j} a[k] C[j][k]) / C[j][j]
template <int M>
__device__ __forceinline__ void csolve_right_lt(double (&ar)[M], double (&ai)[M], const double (&cr)[M], const double (&ci)[M],
                                                const double (&rd)[M]) {
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        sfor<0, j>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<j>(ar[j], cr[k], ar[k]);
            fmac_bc<j>(ar[j], ci[k], ai[k]);
            fnmac_bc<j>(ai[j], ci[k], ar[k]);
            fnmac_bc<j>(ai[j], cr[k], ai[k]);
        });
        ar[j] *= rd[j];
        ai[j] *= rd[j];
    });
}

// Complex Householder tridiagonalisation of the Hermitian H held one row per lane (hr[j] + i hi[j] = H[me][j]).
// d_k and |b_k|^2 are group-uniform; the lane with keep = true stores them.
//
// TB >= 2: only the first M - TB steps run here.  The TB x TB Hermitian block that is left is PARKED in the LDS, in the
// column of the lane that keeps the pair (column `keeper` of `park_all`: element e of the block at park_all[65 e + keeper]),
// and tridiagonalised one pair per lane after the last round (siegel_math.hpp herm_tridiagonalize): a step in this layout
// costs ~110 wave instructions of group-uniform work next to its 12 (M - k - 1) DPP FMAs and serves four pairs, the
// same step one pair per lane serves 64 (the hybrid of the spd forward kernel, spd_coop.hpp; there the block travels in
// registers, here the registers are taken and the LDS is not).  Block layout: the upper triangle of Re H row by row
// (element (i, j), i <= j, at park_slot(i, j)), then the same of Im H.
template <int TB>
constexpr int park_base(const int i) { return i * TB - i * (i + 1) / 2; }      // park_slot(i, j) = park_base(i) + j
template <int TB>
constexpr int park_slot(const int i, const int j) { return park_base<TB>(i) + j; }
template <int TB>
constexpr int park_doubles() { return TB * (TB + 1); }                          // per pair, both planes
constexpr int PARK_STRIDE = 65;     // doubles between two elements: one column per lane of the wave + a dummy one

template <int M, int TB = 0>
__device__ __forceinline__ void tridiagonalize_rows(double (&hr)[M], double (&hi)[M], const int r, const bool keep,
                                                    double (&d)[M], double (&e2)[M], double* __restrict__ park_all = nullptr,
                                                    const int keeper = 0) {
    constexpr int KS = (TB >= 2) ? M - TB : M - 2;
    sfor<0, KS>([&](auto K) {
        constexpr int k = K;
        // my element of column k: H[me][k]; every use of the reflector broadcasts from these registers
        const double cr = settle(hr[k]), ci = settle(hi[k]);
        const double x0r = bcast<k + 1>(cr), x0i = bcast<k + 1>(ci);
        const double dk = bcast<k>(cr);
        const double t2 = (r > k + 1 && r < M) ? sympa::d_fma(cr, cr, ci * ci) : 0.0;
        const double s2 = group_sum(t2);
        const double x02 = sympa::d_fma(x0r, x0r, x0i * x0i);
        const double n2 = x02 + s2;
        d[k] = keep ? dk : d[k];
        e2[k] = keep ? n2 : e2[k];
        const double nx = sympa::d_sqrt(n2);
        const double ix0 = sympa::d_rsqrt(x02 + sympa::TINY);
        const double ax0 = x02 * ix0;
        const bool x0zero = !(x02 > 0.0);
        const double pr = x0zero ? 1.0 : x0r * ix0, pi = x0zero ? 0.0 : x0i * ix0;      // phase of x0
        const double v0r = pr * (ax0 + nx), v0i = pi * (ax0 + nx);                      // v = x + phase ||x|| e1
        const double beta = (s2 > 0.0) ? sympa::d_rcp(nx * (nx + ax0)) : 0.0;           // 2 / ||v||^2
        const double vr = settle((r <= k || r >= M) ? 0.0 : ((r == k + 1) ? v0r : cr));
        const double vi = settle((r <= k || r >= M) ? 0.0 : ((r == k + 1) ? v0i : ci));
        // p = beta H v:  p_i = sum_j H[i][j] v_j
        double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
        sfor<k + 1, M>([&](auto J) {
            constexpr int j = J;
            fmac_bc<j>(p0, vr, hr[j]);
            fnmac_bc<j>(p1, vi, hi[j]);
            fmac_bc<j>(q0, vi, hr[j]);
            fmac_bc<j>(q1, vr, hi[j]);
        });
        double pr_ = (r <= k || r >= M) ? 0.0 : beta * (p0 + p1);
        double pi_ = (r <= k || r >= M) ? 0.0 : beta * (q0 + q1);
        const double kk = 0.5 * beta * group_sum(sympa::d_fma(vr, pr_, vi * pi_));       // Re(v^H p) beta / 2
        const double qr = settle(sympa::d_fma(-kk, vr, pr_));
        const double qi = settle(sympa::d_fma(-kk, vi, pi_));
        // H <- H - v q^H - q v^H:   H[i][j] -= v_i conj(q_j) + q_i conj(v_j)
        sfor<k + 1, M>([&](auto J) {
            constexpr int j = J;
            fnmac_bc<j>(hr[j], qr, vr);
            fnmac_bc<j>(hr[j], qi, vi);
            fnmac_bc<j>(hr[j], vr, qr);
            fnmac_bc<j>(hr[j], vi, qi);
            fnmac_bc<j>(hi[j], qr, vi);
            fmac_bc<j>(hi[j], qi, vr);
            fnmac_bc<j>(hi[j], vr, qi);
            fmac_bc<j>(hi[j], vi, qr);
        });
    });
    if constexpr (TB >= 2) {
        // Row i of the block (lane KS + i) owns slots park_base(i) + j, j >= i.  No branch per row: every lane of the block
        // stores ALL its TB columns from its own base; what lane i writes for j < i lands in slots of the rows above it,
        // whose owners write them with a LATER instruction (their column index is larger by park_base(i) - park_base(i') > 0),
        // and the LDS executes a wave's stores in program order.
        // (No branch around the stores either: the lanes outside the block store their junk into a 65th, dummy column.)
        {
            const bool in_block = (r >= KS && r < M);
            const int i = in_block ? r - KS : 0;
            double* const row = park_all + PARK_STRIDE * (i * TB - i * (i + 1) / 2) + (in_block ? keeper : 64);
            sfor<0, TB>([&](auto J) {
                row[PARK_STRIDE * J] = hr[KS + J];
                row[PARK_STRIDE * (TB * (TB + 1) / 2 + J)] = hi[KS + J];
                // column J before column J + 1, as separate instructions: to the compiler a lane's stores go to different
                // addresses and may be reordered or paired into one ds_write2 (measured: TB = 7 wrong without this)
                asm volatile("" ::: "memory");
            });
        }
    } else {
        const double lr = settle(hr[M - 1]), li = settle(hi[M - 1]);
        const double dm = bcast<M - 2>(settle(hr[M - 2]));
        const double dn = bcast<M - 1>(lr);
        const double br = bcast<M - 2>(lr), bi = bcast<M - 2>(li);
        d[M - 2] = keep ? dm : d[M - 2];
        d[M - 1] = keep ? dn : d[M - 1];
        e2[M - 2] = keep ? sympa::d_fma(br, br, bi * bi) : e2[M - 2];
    }
}

// The parked block of MY pair (column `lane` of the park) tridiagonalised: d[KS ..], e2[KS ..] of the pair's form.
template <int M, int TB>
__device__ __forceinline__ void finish_parked(const double* __restrict__ park_all, const int lane, double (&d)[M], double (&e2)[M]) {
    sympa::Herm<TB> hb;
    constexpr int IM = TB * (TB + 1) / 2;
    sfor<0, TB>([&](auto I) {
        constexpr int i = I;
        hb.d[i] = park_all[PARK_STRIDE * park_slot<TB>(i, i) + lane];
        sfor<i + 1, TB>([&](auto J) {
            constexpr int j = J;
            hb.re[i][j] = park_all[PARK_STRIDE * park_slot<TB>(i, j) + lane];
            hb.im[i][j] = park_all[PARK_STRIDE * (IM + park_slot<TB>(i, j)) + lane];
        });
    });
    double aa[TB], bb[TB];
    sympa::herm_tridiagonalize<TB>(hb, aa, bb);
    sfor<0, TB>([&](auto I) { d[M - TB + I] = aa[I]; e2[M - TB + I] = bb[I]; });
}

}  // namespace siegel_coop
