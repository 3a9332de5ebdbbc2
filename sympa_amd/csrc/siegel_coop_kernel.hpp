// Kernel template of the upper / bounded models for 9 <= n <= 16: sixteen lanes per pair (siegel_coop.hpp), instantiated
// per matrix size M = n by siegel_coop_upper.hip / siegel_coop_bounded.hip (one translation unit per model: the sixteen
// fully unrolled kernels compile in parallel).
#pragma once
#include "siegel_common.hpp"
#include "siegel_coop.hpp"

namespace sympa_hip {

// One wave per block, 64 pairs per wave in 16 rounds of 4; lane 16 g + t owns pair 4 t + g (see spd.hip).
// Lanes r >= M of a group are phantoms (spd_coop.hpp): they run the same instructions on values nobody reads.
// Small M: the working set fits 256 registers once the next round's rows are not prefetched, so two waves
// share a SIMD and hide each other's load latency (measured per M, profiles/r02_dims_sweep.txt); larger M: one
// 512-register wave per SIMD that prefetches the rows of round t + 1 while round t computes.
// (upper: M <= 12 needs 230 registers, M = 13 exactly 256, M = 14 spills 20 and is still 10 % faster than one wave per
// SIMD, M = 15 breaks even, M = 16 loses 3 %; bounded, whose factors are complex: M <= 10 needs 254, M = 11 spills 27
// and gains 6 %, M = 12 spills 66 and loses 9 % -- profiles/r03_coop_two_waves.txt)
template <int MODEL, int M>
constexpr bool coop_two_waves() { return M <= (MODEL == sympa::MODEL_UPPER ? 14 : 11); }
// Pairs per wave and round / rounds per wave: 4 and 16 with sixteen lanes per pair; 8 and 8 in a unit that defines
// SYMPA_COOP_HALF (eight lanes per pair, M <= 8; siegel_coop_half.hip: the A/B of the dims 7, 8 forward against the
// one-pair-per-lane kernels).  Lane GROUP g + t owns pair GPW t + g of the wave in the one-pair-per-lane QL phase.
// Size of the trailing block that is parked in the LDS and tridiagonalised one pair per lane (siegel_coop.hpp): what the
// LDS of the occupancy allows -- 20 KB per wave at two waves per SIMD (4 x 4: 10.4 KB next to the 9.7 KB of the transposition
// buffers), 40 KB at one (7 x 7: 29.1 KB).  Eight lanes per pair (A/B unit): none.
constexpr int SIEGEL_TB_TWO_WAVES = 4;
constexpr int SIEGEL_TB_ONE_WAVE = 7;
// (The stores of the block carry no branch: with one `if (lane is in the block)` around them the two-wave variants needed
// ~20 registers more and upper M = 12..14 spilled and lost 0-19 %; with every lane storing -- the outside lanes into a dummy
// column -- they need FEWER registers than without the parked block: profiles/r03_siegel_parked_block.txt.)
template <int MODEL, int M>
constexpr int coop_parked_block() {
    return spd_coop::GROUP != 16 ? 0 : (coop_two_waves<MODEL, M>() ? SIEGEL_TB_TWO_WAVES : SIEGEL_TB_ONE_WAVE);
}
constexpr int COOP_GPW = spd_coop::GROUPS_PER_WAVE;
constexpr int COOP_ROUNDS = 64 / COOP_GPW;

template <int MODEL, int M>
__global__ __launch_bounds__(64, (coop_two_waves<MODEL, M>() ? 2 : 1)) void siegel_coop_kernel(const DistArgs a) {
    constexpr bool PREFETCH = !coop_two_waves<MODEL, M>();
    using namespace siegel_coop;
    constexpr int TB = coop_parked_block<MODEL, M>();
    static_assert(TB == 0 || (TB >= 2 && TB <= M - 2), "parked block");
    __shared__ __attribute__((aligned(16))) double tbuf_all[COOP_GPW * spd_coop::TBUF];
    __shared__ double park_all[TB >= 2 ? TB * (TB + 1) * siegel_coop::PARK_STRIDE : 1];
    static_assert(M <= spd_coop::GROUP, "matrix rows per group");
    const int lane = threadIdx.x;
    const int g = lane / spd_coop::GROUP, r = lane % spd_coop::GROUP;
    const int64_t i = (int64_t)blockIdx.x * 64 + COOP_GPW * r + g;
    const bool live = i < a.b;
    const int64_t ii = live ? i : a.b - 1;
    int st = 0;
    int64_t r1 = ii, r2 = ii;
    if (a.ap_cols > 0) {
        ap_pair(a, ii, r1, r2);
    } else if (a.idx1 != nullptr) {
        r1 = a.idx1[ii * a.idx1_stride];
        r2 = a.idx2[ii * a.idx2_stride];
        if (r1 < 0 || r1 >= a.num_rows || r2 < 0 || r2 >= a.num_rows) { st |= sympa::ST_BAD_INDEX; r1 = 0; r2 = 0; }
    }
    const int row1 = (int)r1, row2 = (int)r2;      // num_rows < 2^31 (checked on the host)
    double* const tbuf = tbuf_all + g * spd_coop::TBUF;
    constexpr int nn = M * M;

    double d[M], e2[M];
#pragma unroll
    for (int k = 0; k < M; ++k) { d[k] = 0.0; e2[k] = 0.0; }
    bool ok = true;
    // rows of round t + 1 are fetched while round t computes (raw elements kept in registers: nothing else hides the
    // latency of the 4 M loads); a phantom lane reads element 0 of its pair's rows
    double fa[M], fb[M], fc[M], fd[M];
    auto fetch = [&](const int t) {
        const int ra = __builtin_amdgcn_ds_bpermute(4 * (spd_coop::GROUP * g + t), row1);     // the rows of my group's pair
        const int rb = __builtin_amdgcn_ds_bpermute(4 * (spd_coop::GROUP * g + t), row2);
        const double* pa = a.base1 + (size_t)(unsigned)ra * (size_t)(2 * nn);
        const double* pb = a.base2 + (size_t)(unsigned)rb * (size_t)(2 * nn);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int lo = r < j ? r : j, hi = r < j ? j : r;
            const int e = (hi < M) ? lo * M + hi : 0;      // upper triangle only: element (min, max)
            fa[j] = pa[e]; fb[j] = pa[nn + e]; fc[j] = pb[e]; fd[j] = pb[nn + e];
        }
    };
    if constexpr (PREFETCH) fetch(0);
    for (int t = 0; t < COOP_ROUNDS; ++t) {
        double er[M], ei[M];
        bool pd1, pd2;
        if constexpr (!PREFETCH) fetch(t);
        if constexpr (MODEL == sympa::MODEL_UPPER) {
            // my row of X1, Y1, X2, Y2
            double dr[M], di[M], y1[M], y2[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                dr[j] = fc[j] - fa[j];
                di[j] = fd[j] - fb[j];
                y1[j] = fb[j];
                y2[j] = fd[j];
            }
            if constexpr (PREFETCH) if (t + 1 < COOP_ROUNDS) fetch(t + 1);
            double rd1[M], rd2[M];
            pd1 = spd_coop::cholesky_rows(y1, rd1);
            pd2 = spd_coop::cholesky_rows(y2, rd2);
            // W = D L2^-T (both planes), E^T = W^T L1^-T
            spd_coop::solve_right_lt(dr, y2, rd2);
            spd_coop::solve_right_lt(di, y2, rd2);
            spd_coop::transpose_rows(dr, er, tbuf, r);
            spd_coop::transpose_rows(di, ei, tbuf, r);
            spd_coop::solve_right_lt(er, y1, rd1);
            spd_coop::solve_right_lt(ei, y1, rd1);
        } else {
            // my row of W1, W2 (Re, Im).  E = C1^-1 (W2 - W1) C2^-T,  I - W_k W_k^H = C_k C_k^H
            double w1r[M], w1i[M], w2r[M], w2i[M];
#pragma unroll
            for (int j = 0; j < M; ++j) { w1r[j] = fa[j]; w1i[j] = fb[j]; w2r[j] = fc[j]; w2i[j] = fd[j]; }
            if constexpr (PREFETCH) if (t + 1 < COOP_ROUNDS) fetch(t + 1);
            double dr[M], di[M];
#pragma unroll
            for (int j = 0; j < M; ++j) { dr[j] = w2r[j] - w1r[j]; di[j] = w2i[j] - w1i[j]; }
            double c2r[M], c2i[M], rd2[M];
            id_minus_wwh_rows(w2r, w2i, c2r, c2i, r);
            pd2 = ccholesky_rows(c2r, c2i, rd2);
            csolve_right_lt(dr, di, c2r, c2i, rd2);              // W = D C2^-T
            spd_coop::transpose_rows(dr, er, tbuf, r);
            spd_coop::transpose_rows(di, ei, tbuf, r);
            double c1r[M], c1i[M], rd1[M];
            id_minus_wwh_rows(w1r, w1i, c1r, c1i, r);
            pd1 = ccholesky_rows(c1r, c1i, rd1);
            csolve_right_lt(er, ei, c1r, c1i, rd1);              // E^T = W^T C1^-T
        }
        double hr[M], hi[M];
        gram_columns(er, ei, hr, hi);
        const bool keep = (r == t);
        ok = keep ? (pd1 && pd2) : ok;
        tridiagonalize_rows<M, TB>(hr, hi, r, keep, d, e2, park_all, spd_coop::GROUP * g + t);
    }
    if constexpr (TB >= 2) {
        wave_lds_fence();
        finish_parked<M, TB>(park_all, lane, d, e2);
    }
    // one pair per lane: eigenvalues of H = E^H E, vector-valued distance, metric
    const bool conv = sympa::tridiag_ql_lockstep<M, 0, true>(d, e2);      // (forward: eigenvalues only)
    double v[M];
    constexpr double quarter = (MODEL == sympa::MODEL_UPPER) ? 0.25 : 1.0;      // sinh(v/2) = sigma / 2 (upper), sigma (bounded)
    bool finite = true;          // tested before the clamp: fmax would turn a NaN eigenvalue into distance 0
#pragma unroll
    for (int k = 0; k < M; ++k) {
        finite = finite && sympa::d_finite(d[k]);
        v[k] = sympa::vvd_from_sinh2(fmax(d[k], 0.0) * quarter, a.inv_eps);
    }
    if (!finite) {
#pragma unroll
        for (int k = 0; k < M; ++k) v[k] = __builtin_nan("");
    }
    if (a.vvd != nullptr) {
        sympa::sort_ascending<M>(v);
        if (live) {
#pragma unroll
            for (int k = 0; k < M; ++k) a.vvd[i * M + k] = v[k];
        }
    }
    double out = sympa::reduce_metric<M>(v, a.metric, a.metric_w);
    if (!finite) out = __builtin_nan("");
    if (!ok) st |= sympa::ST_NOT_PD;
    if (!conv) st |= sympa::ST_NO_CONVERGENCE;
    if (!sympa::d_finite(out)) st |= sympa::ST_NONFINITE;
    if (st & sympa::ST_BAD_INDEX) out = __builtin_nan("");
    if (a.scale != nullptr) out *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);
    if (live) {
        if (a.ap_cols > 0) ap_store(a, i, r1, r2, out);
        else a.out[i] = out;
    }
    if (a.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long mk = __ballot(flagged);
        if (mk != 0ull) {
            if (flagged) atomicOr(&a.status[0], st);
            if (lane == 0) atomicAdd(&a.status[1], (int)__popcll(mk));
        }
    }
}

template <int MODEL, int M>
int launch_siegel_coop_m(const DistArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((siegel_coop_kernel<MODEL, M>), dim3((unsigned)((a.b + 63) / 64)), dim3(64), 0, s, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

template <int MODEL>
int launch_siegel_coop_model(const DistArgs& a, int n, hipStream_t s) {
    switch (n) {
        case 6: return launch_siegel_coop_m<MODEL, 6>(a, s);     // A/B against the one-pair-per-lane kernels (SYMPA_FLAG_COOP)
        case 8: return launch_siegel_coop_m<MODEL, 8>(a, s);
        case 9: return launch_siegel_coop_m<MODEL, 9>(a, s);
        case 10: return launch_siegel_coop_m<MODEL, 10>(a, s);
        case 11: return launch_siegel_coop_m<MODEL, 11>(a, s);
        case 12: return launch_siegel_coop_m<MODEL, 12>(a, s);
        case 13: return launch_siegel_coop_m<MODEL, 13>(a, s);
        case 14: return launch_siegel_coop_m<MODEL, 14>(a, s);
        case 15: return launch_siegel_coop_m<MODEL, 15>(a, s);
        case 16: return launch_siegel_coop_m<MODEL, 16>(a, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "sixteen-lanes-per-pair kernels cover dims 9..16");
    }
}

int launch_siegel_coop_upper(const DistArgs& a, int n, hipStream_t s);
int launch_siegel_coop_bounded(const DistArgs& a, int n, hipStream_t s);

}  // namespace sympa_hip
