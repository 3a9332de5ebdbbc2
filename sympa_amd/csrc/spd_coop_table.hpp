// SPD model, optimiser-side row operations with sixteen lanes per table row (layout and DPP machinery of spd_coop.hpp):
//   OP_EGRAD2RGRAD  out = x sym(g) x                                      (geoopt SymmetricPositiveDefinite.egrad2rgrad)
//   OP_RSGD         x <- sym(x + u + 1/2 u x^-1 u),  u = -lr x sym(coef g + wd x) x          (RiemannianSGD step, retr)
// Lane r of a group owns row r of x, of the gradient and of every product:
//   (A B)[me][j] = sum_k A[me][k] B[k][j]           one fmac_dpp per term, B[k][j] = register j of lane k
//   u x^-1 u = W^T W with W = L^-1 u: rows of W^T = u L^-T by solve_right_lt (u symmetric), then
//   (W^T W)[me][j] = sum_k W^T[me][k] W^T[j][k]     my register k times lane j's register k
// Same formulas as spd_row_egrad2rgrad / spd_row_rsgd (spd_math_bwd.hpp, one row per lane in scratch memory: 6.3 ms per
// 100 000 rows at n = 16); the tests check both against the g++ build of those and against the oracle.
#pragma once

#include "spd_coop.hpp"

namespace spd_coop {

constexpr int OP_PROJX = 0, OP_RSGD = 1, OP_EGRAD2RGRAD = 2, OP_SQNORM = 3;      // the op numbers of siegel_table.hip

// t = a b for the rows held one per lane (b's rows are broadcast)
template <int M>
__device__ __forceinline__ void matmul_rows(const double (&a)[M], double (&b)[M], double (&t)[M]) {
    sfor<0, M>([&](auto J) { b[J] = settle(b[J]); });
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        double a0 = 0.0, a1 = 0.0;
        sfor<0, M>([&](auto K) {
            constexpr int k = K;
            if constexpr (k % 2 == 0) fmac_bc<k>(a0, b[j], a[k]);
            else fmac_bc<k>(a1, b[j], a[k]);
        });
        t[j] = a0 + a1;
    });
}

template <int M, int OP>
__global__ __launch_bounds__(64) void spd_coop_table_kernel(double* __restrict__ x, const double* __restrict__ g,
                                                            double* __restrict__ out, const int64_t b, const double lr,
                                                            const double wd, const double* __restrict__ clip,
                                                            const double max_norm, int32_t* __restrict__ status, const int rounds) {
    __shared__ __attribute__((aligned(16))) double tbuf_all[4 * TBUF];
    const int lane = threadIdx.x;
    const int grp = lane >> 4, r = lane & 15;
    double* const tbuf = tbuf_all + grp * TBUF;
    constexpr int nn = M * M;
    const double coef = (clip != nullptr) ? fmin(1.0, max_norm / (sqrt(clip[0]) + 1e-6)) : 1.0;
    int st = 0, nbad = 0;
    for (int t = 0; t < rounds; ++t) {
        const int64_t first = ((int64_t)blockIdx.x * rounds + t) * 4;
        if (first >= b) break;                                   // wave-uniform
        const int64_t i = first + grp;
        const bool live = i < b;
        const int64_t ii = live ? i : b - 1;
        const double* px = x + ii * nn;
        const double* pg = g + ii * nn;
        const int rr = r < M ? r : 0;                       // a phantom lane reads row 0
        double xs[M], u[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            xs[j] = 0.5 * (px[rr * M + j] + px[j * M + rr]);
            u[j] = 0.5 * (pg[rr * M + j] + pg[j * M + rr]);
        }
        double tt[M], p[M];
        if constexpr (OP == OP_EGRAD2RGRAD) {
            matmul_rows(xs, u, tt);
            double xb[M];
#pragma unroll
            for (int j = 0; j < M; ++j) xb[j] = xs[j];
            matmul_rows(tt, xb, p);
            if (live && r < M) {
#pragma unroll
                for (int j = 0; j < M; ++j) out[i * nn + r * M + j] = p[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < M; ++j) u[j] = sympa::d_fma(wd, xs[j], coef * u[j]);
            matmul_rows(xs, u, tt);
            double l[M], rd[M];
#pragma unroll
            for (int j = 0; j < M; ++j) l[j] = xs[j];
            matmul_rows(tt, l, p);                          // x sym(.) x  (l = x, settled by matmul_rows)
            // symmetrise u = -lr p through the transpose
            transpose_rows(p, tt, tbuf, r);
#pragma unroll
            for (int j = 0; j < M; ++j) u[j] = -0.5 * lr * (p[j] + tt[j]);
            const bool pd = cholesky_rows(l, rd);
#pragma unroll
            for (int j = 0; j < M; ++j) tt[j] = u[j];
            solve_right_lt(tt, l, rd);                      // rows of W^T = u L^-T
#pragma unroll
            for (int j = 0; j < M; ++j) tt[j] = settle(tt[j]);
            if (live && r < M) {
                double* po = x + i * nn + r * M;
                sfor<0, M>([&](auto J) {
                    constexpr int j = J;
                    double a0 = 0.0, a1 = 0.0;
                    sfor<0, M>([&](auto K) {
                        constexpr int k = K;
                        if constexpr (k % 2 == 0) fmac_bc<j>(a0, tt[k], tt[k]);
                        else fmac_bc<j>(a1, tt[k], tt[k]);
                    });
                    po[j] = xs[j] + u[j] + 0.5 * (a0 + a1);
                });
            }
            if (live && !pd && r == 0) { st |= sympa::ST_NOT_PD; ++nbad; }
        }
    }
    if (status != nullptr) {
        if (__ballot(st != 0) != 0ull) {
            if (st != 0) atomicOr(&status[0], st);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nbad += __shfl_xor(nbad, off);
            if (lane == 0) atomicAdd(&status[1], nbad);          // rows flagged
        }
    }
}

}  // namespace spd_coop
