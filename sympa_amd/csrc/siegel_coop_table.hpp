// Upper-half / bounded models, optimiser-side row operations for 9 <= n <= 16 with sixteen lanes per table row (layout and
// DPP machinery of spd_coop.hpp): egrad2rgrad, the RiemannianSGD step, projx and the squared tangent norm (inner).  The one-row-per-lane kernels of these dims
// (siegel_table_rolled.hip, per-lane scratch) take 2.4 ms (n = 10) to 10-19 ms (n = 16) per step over 5 041 rows --
// more than the backward of a 65 536-pair batch.
//   upper  : egrad2rgrad = Y G Y on both planes (four real row-broadcast products)          upper_half.py:25-40
//   bounded: egrad2rgrad = A G A,  A = I - conj(Z) Z (three complex products)               bounded_domain.py:41-53
//   step   : z <- projx(z - lr * egrad2rgrad(z, grad + wd z)),  projx = symmetrise, then clamp the spectrum
//            (upper_half.py:42-66, bounded_domain.py:55-84, siegel_manifold.py:74-87)
// The clamp only acts on a point that left the manifold's eps-interior, which a training step rarely does.  This kernel
// symmetrises through a transpose, TESTS "inside" with a Cholesky factorisation --
//   upper  : all eigenvalues of Y > eps         <=>  Y - eps I positive definite
//   bounded: all Takagi values of Z < 1 - eps   <=>  I - W W^H positive definite, W = Z / (1 - eps)
// -- writes the symmetrised row, and counts the rows that fail the test in a device word; the host then launches the
// one-row-per-lane projx over the table gated on that word (it returns at once when the count is zero; otherwise it
// applies the reference's exact eigenvalue clamp, which leaves inside rows untouched).
#pragma once

#include "siegel_coop_bwd.hpp"
#include "spd_coop_table.hpp"

namespace siegel_coop {

template <int MODEL, int M, int OP>
__global__ __launch_bounds__(64) void siegel_coop_table_kernel(double* __restrict__ z, const double* __restrict__ grad,
                                                               double* __restrict__ out, const int64_t b, const double lr,
                                                               const double wd, const double eps,
                                                               const double* __restrict__ clip, const double max_norm,
                                                               int* __restrict__ outside, const int rounds) {
    constexpr bool UPPER = (MODEL == sympa::MODEL_UPPER);
    using spd_coop::matmul_rows;
    using spd_coop::transpose_rows;
    __shared__ __attribute__((aligned(16))) double tbuf_all[spd_coop::GROUPS_PER_WAVE * spd_coop::TBUF];
    const int lane = threadIdx.x;
    const int grp = lane / spd_coop::GROUP, r = lane % spd_coop::GROUP;
    double* const tbuf = tbuf_all + grp * spd_coop::TBUF;
    constexpr int nn = M * M;
    constexpr int64_t ROW = 2 * nn;
    const double coef = (clip != nullptr) ? fmin(1.0, max_norm / (sqrt(clip[0]) + 1e-6)) : 1.0;
    int nout = 0;
    for (int t = 0; t < rounds; ++t) {
        const int64_t first = ((int64_t)blockIdx.x * rounds + t) * spd_coop::GROUPS_PER_WAVE;
        if (first >= b) break;                                   // wave-uniform
        const int64_t i = first + grp;
        const bool live = i < b;
        const int64_t ii = live ? i : b - 1;
        const int rr = r < M ? r : 0;                            // a phantom lane reads row 0
        const double* pz = z + ii * ROW + rr * M;
        double zr[M], zi[M], gr[M], gi[M];
#pragma unroll
        for (int j = 0; j < M; ++j) { zr[j] = pz[j]; zi[j] = pz[nn + j]; }
        if constexpr (OP != spd_coop::OP_PROJX) {
            const double* pg = grad + ii * ROW + rr * M;
#pragma unroll
            for (int j = 0; j < M; ++j) { gr[j] = pg[j]; gi[j] = pg[nn + j]; }
        }
        if constexpr (OP == spd_coop::OP_SQNORM) {
            // inner(z, u, u) = Re tr[Q conj(Q)] = sum_ij (Re q_ij Re q_ji + Im q_ij Im q_ji)   (siegel_table_math.hpp)
            //   upper: Q = L^-1 u L^-T, Y = L L^T;   bounded: Q = C^-1 conj(u) C^-T, I - Z Z^H = C C^H
            double qr[M], qi[M], tr[M], ti[M];
            bool pd;
            if constexpr (UPPER) {
                double l[M], rd[M];
#pragma unroll
                for (int j = 0; j < M; ++j) l[j] = zi[j];
                pd = spd_coop::cholesky_rows(l, rd);
                spd_coop::solve_right_lt2(gr, gi, l, rd);                 // u L^-T
                transpose_rows(gr, qr, tbuf, r);
                transpose_rows(gi, qi, tbuf, r);
                spd_coop::solve_right_lt2(qr, qi, l, rd);                 // rows of (L^-1 u L^-T)^T
            } else {
                double cr[M], ci[M], rd[M];
                id_minus_wwh_rows(zr, zi, cr, ci, r);
                pd = ccholesky_rows(cr, ci, rd);
#pragma unroll
                for (int j = 0; j < M; ++j) gi[j] = -gi[j];               // conj(u)
                csolve_right_lt(gr, gi, cr, ci, rd);
                transpose_rows(gr, qr, tbuf, r);
                transpose_rows(gi, qi, tbuf, r);
                csolve_right_lt(qr, qi, cr, ci, rd);                      // rows of Q^T
            }
            transpose_rows(qr, tr, tbuf, r);
            transpose_rows(qi, ti, tbuf, r);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) acc = sympa::d_fma(qr[j], tr[j], sympa::d_fma(qi[j], ti[j], acc));
            acc = group_sum((r < M) ? acc : 0.0);
            if (live && r == 0) out[i] = acc;
            if (live && !pd && r == 0) ++nout;                            // reported as "not positive definite" by the host side
            continue;
        }

        if constexpr (OP == spd_coop::OP_RSGD) {
#pragma unroll
            for (int j = 0; j < M; ++j) { gr[j] = sympa::d_fma(wd, zr[j], coef * gr[j]); gi[j] = sympa::d_fma(wd, zi[j], coef * gi[j]); }
        }
        // (rr, ri) = my row of egrad2rgrad(z, g)
        double rgr[M], rgi[M];
        if constexpr (OP == spd_coop::OP_PROJX) {
            // nothing to add: projx(z) alone
        } else if constexpr (UPPER) {
            double y[M], tt[M];
#pragma unroll
            for (int j = 0; j < M; ++j) y[j] = zi[j];
            matmul_rows(zi, gr, tt);                             // Y Gr
            matmul_rows(tt, y, rgr);                             // (Y Gr) Y
            matmul_rows(zi, gi, tt);
            matmul_rows(tt, y, rgi);
        } else {
            double ar[M], ai[M], tr[M], ti[M], nzi[M], wr[M], wi[M];
#pragma unroll
            for (int j = 0; j < M; ++j) { nzi[j] = -zi[j]; wr[j] = zr[j]; wi[j] = zi[j]; }
            cmatmul_rows(zr, nzi, wr, wi, ar, ai);               // conj(Z) Z
#pragma unroll
            for (int j = 0; j < M; ++j) { ar[j] = ((r == j) ? 1.0 : 0.0) - ar[j]; ai[j] = -ai[j]; }   // A = I - conj(Z) Z
            cmatmul_rows(ar, ai, gr, gi, tr, ti);                // A G
            double br[M], bi[M];
#pragma unroll
            for (int j = 0; j < M; ++j) { br[j] = ar[j]; bi[j] = ai[j]; }
            cmatmul_rows(tr, ti, br, bi, rgr, rgi);              // (A G) A
        }
        if constexpr (OP == spd_coop::OP_EGRAD2RGRAD) {
            if (live && r < M) {
                double* po = out + i * ROW + r * M;
#pragma unroll
                for (int j = 0; j < M; ++j) { po[j] = rgr[j]; po[nn + j] = rgi[j]; }
            }
        } else {
            // x + u, symmetrised through the transpose
            double tr[M], ti[M];
            if constexpr (OP == spd_coop::OP_RSGD) {
#pragma unroll
                for (int j = 0; j < M; ++j) { zr[j] = sympa::d_fma(-lr, rgr[j], zr[j]); zi[j] = sympa::d_fma(-lr, rgi[j], zi[j]); }
            }
            transpose_rows(zr, tr, tbuf, r);
            transpose_rows(zi, ti, tbuf, r);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const double sr = 0.5 * (zr[j] + tr[j]), si = 0.5 * (zi[j] + ti[j]);
                zr[j] = (r == j) ? zr[j] : sr;                   // the diagonal is left as it is (symmetrise(), siegel_table_math.hpp)
                zi[j] = (r == j) ? zi[j] : si;
            }
            if (live && r < M) {
                double* po = (OP == spd_coop::OP_PROJX ? out : z) + i * ROW + r * M;
#pragma unroll
                for (int j = 0; j < M; ++j) { po[j] = zr[j]; po[nn + j] = zi[j]; }
            }
            // inside the eps-interior?
            bool pd;
            if constexpr (UPPER) {
                double c[M], rd[M];
#pragma unroll
                for (int j = 0; j < M; ++j) c[j] = zi[j] - ((r == j) ? eps : 0.0);
                pd = spd_coop::cholesky_rows(c, rd);
            } else {
                const double s = 1.0 / (1.0 - eps);
                double wr[M], wi[M], cr[M], ci[M], rd[M];
#pragma unroll
                for (int j = 0; j < M; ++j) { wr[j] = s * zr[j]; wi[j] = s * zi[j]; }
                id_minus_wwh_rows(wr, wi, cr, ci, r);
                pd = ccholesky_rows(cr, ci, rd);
            }
            if (live && !pd && r == 0) ++nout;
        }
    }
    if constexpr (OP != spd_coop::OP_EGRAD2RGRAD) {
        if (outside != nullptr && __ballot(nout != 0) != 0ull) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nout += __shfl_xor(nout, off);
            if (lane == 0) {
                if constexpr (OP == spd_coop::OP_SQNORM) {       // `outside` is the status pair here: a point off the manifold
                    atomicOr(&outside[0], sympa::ST_NOT_PD);
                    atomicAdd(&outside[1], nout);
                } else {
                    atomicAdd(outside, nout);                    // rows that left the eps-interior
                }
            }
        }
    }
}

}  // namespace siegel_coop
