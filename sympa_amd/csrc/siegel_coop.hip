// gfx950 kernels of the upper and bounded models for 9 <= n <= 16: sixteen lanes per pair (siegel_coop.hpp).
#include "siegel_common.hpp"
#include "siegel_coop.hpp"

namespace sympa_hip {
namespace {

// One wave per block, 64 pairs per wave in 16 rounds of 4; lane 16 g + t owns pair 4 t + g (see spd.hip).
template <int MODEL>
__global__ __launch_bounds__(64) void siegel_coop_kernel(const DistArgs a, const int n) {
    using namespace siegel_coop;
    __shared__ __attribute__((aligned(16))) double tbuf_all[4 * N * N];
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15;
    const int64_t i = (int64_t)blockIdx.x * 64 + 4 * r + g;
    const bool live = i < a.b;
    const int64_t ii = live ? i : a.b - 1;
    int st = 0;
    int64_t r1 = ii, r2 = ii;
    if (a.ap_cols > 0) {
        r1 = a.ap_row0 + ii / a.ap_cols;
        r2 = ii % a.ap_cols;
    } else if (a.idx1 != nullptr) {
        r1 = a.idx1[ii * a.idx1_stride];
        r2 = a.idx2[ii * a.idx2_stride];
        if (r1 < 0 || r1 >= a.num_rows || r2 < 0 || r2 >= a.num_rows) { st |= sympa::ST_BAD_INDEX; r1 = 0; r2 = 0; }
    }
    const int row1 = (int)r1, row2 = (int)r2;      // num_rows < 2^31 (checked on the host)
    double* const tbuf = tbuf_all + g * N * N;
    const int nn = n * n;

    double d[N], e2[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { d[k] = 0.0; e2[k] = 0.0; }
    bool ok = true;
    // rows of round t + 1 are fetched while round t computes (raw elements kept in registers: one wave per SIMD, the
    // register file has the room and nothing else hides the latency of the 64 loads)
    double fa[N], fb[N], fc[N], fd[N];
    auto fetch = [&](const int t) {
        const int ra = __builtin_amdgcn_ds_bpermute(4 * (16 * g + t), row1);     // the rows of my group's pair
        const int rb = __builtin_amdgcn_ds_bpermute(4 * (16 * g + t), row2);
        const double* pa = a.base1 + (size_t)(unsigned)ra * (size_t)(2 * nn);
        const double* pb = a.base2 + (size_t)(unsigned)rb * (size_t)(2 * nn);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int lo = r < j ? r : j, hi = r < j ? j : r;
            const int e = (hi < n) ? lo * n + hi : 0;
            fa[j] = pa[e]; fb[j] = pa[nn + e]; fc[j] = pb[e]; fd[j] = pb[nn + e];
        }
    };
    fetch(0);
    for (int t = 0; t < spd_coop::ROUNDS; ++t) {
        double er[N], ei[N];
        bool pd1, pd2;
        if constexpr (MODEL == sympa::MODEL_UPPER) {
            // my row of X1, Y1, X2, Y2 (upper triangle only: element (min, max)); padding = the point i I
            double dr[N], di[N], y1[N], y2[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int hi = r < j ? j : r;
                const bool inside = hi < n;
                const double ident = (r == j) ? 1.0 : 0.0;
                dr[j] = inside ? fc[j] - fa[j] : 0.0;
                di[j] = inside ? fd[j] - fb[j] : 0.0;
                y1[j] = inside ? fb[j] : ident;
                y2[j] = inside ? fd[j] : ident;
            }
            if (t + 1 < spd_coop::ROUNDS) fetch(t + 1);
            double rd1[N], rd2[N];
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
            // my row of W1, W2 (Re, Im); padding = the point 0.  E = C1^-1 (W2 - W1) C2^-T,  I - W_k W_k^H = C_k C_k^H
            double w1r[N], w1i[N], w2r[N], w2i[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int hi = r < j ? j : r;
                const bool inside = hi < n;
                w1r[j] = inside ? fa[j] : 0.0;
                w1i[j] = inside ? fb[j] : 0.0;
                w2r[j] = inside ? fc[j] : 0.0;
                w2i[j] = inside ? fd[j] : 0.0;
            }
            if (t + 1 < spd_coop::ROUNDS) fetch(t + 1);
            double dr[N], di[N];
#pragma unroll
            for (int j = 0; j < N; ++j) { dr[j] = w2r[j] - w1r[j]; di[j] = w2i[j] - w1i[j]; }
            double c2r[N], c2i[N], rd2[N];
            id_minus_wwh_rows(w2r, w2i, c2r, c2i, r);
            pd2 = ccholesky_rows(c2r, c2i, rd2);
            csolve_right_lt(dr, di, c2r, c2i, rd2);              // W = D C2^-T
            spd_coop::transpose_rows(dr, er, tbuf, r);
            spd_coop::transpose_rows(di, ei, tbuf, r);
            double c1r[N], c1i[N], rd1[N];
            id_minus_wwh_rows(w1r, w1i, c1r, c1i, r);
            pd1 = ccholesky_rows(c1r, c1i, rd1);
            csolve_right_lt(er, ei, c1r, c1i, rd1);              // E^T = W^T C1^-T
        }
        double hr[N], hi[N];
        gram_columns(er, ei, hr, hi);
        const bool keep = (r == t);
        ok = keep ? (pd1 && pd2) : ok;
        tridiagonalize_rows(hr, hi, r, keep, d, e2);
    }
    // one pair per lane: eigenvalues of H = E^H E, vector-valued distance, metric
    const bool conv = sympa::tridiag_ql_lockstep<N>(d, e2);
    double v[N];
    constexpr double quarter = (MODEL == sympa::MODEL_UPPER) ? 0.25 : 1.0;      // sinh(v/2) = sigma / 2 (upper), sigma (bounded)
    bool finite = true;          // tested before the clamp: fmax would turn a NaN eigenvalue into distance 0
#pragma unroll
    for (int k = 0; k < N; ++k) {
        finite = finite && sympa::d_finite(d[k]);
        v[k] = sympa::vvd_from_sinh2(fmax(d[k], 0.0) * quarter, a.inv_eps);
    }
    sympa::sort_ascending<N>(v);
    if (!finite) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = __builtin_nan("");
    }
    const int pad = N - n;
    if (a.vvd != nullptr && live) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (k >= pad) a.vvd[i * n + (k - pad)] = v[k];
    }
    double out = reduce_metric_padded(v, n, a.metric, a.metric_w);
    if (!finite) out = __builtin_nan("");
    if (!ok) st |= sympa::ST_NOT_PD;
    if (!conv) st |= sympa::ST_NO_CONVERGENCE;
    if (!(out == out) || !(fabs(out) <= 1.79e308)) st |= sympa::ST_NONFINITE;
    if (st & sympa::ST_BAD_INDEX) out = __builtin_nan("");
    if (a.scale != nullptr) out *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);
    if (live) a.out[i] = out;
    if (a.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long mk = __ballot(flagged);
        if (mk != 0ull) {
            if (flagged) atomicOr(&a.status[0], st);
            if (lane == 0) atomicAdd(&a.status[1], (int)__popcll(mk));
        }
    }
}

}  // namespace

int launch_siegel_coop(const DistArgs& a, int n, int model, hipStream_t s) {
    const dim3 grid((unsigned)((a.b + 63) / 64));
    if (model == SYMPA_MODEL_UPPER) hipLaunchKernelGGL(siegel_coop_kernel<sympa::MODEL_UPPER>, grid, dim3(64), 0, s, a, n);
    else hipLaunchKernelGGL(siegel_coop_kernel<sympa::MODEL_BOUNDED>, grid, dim3(64), 0, s, a, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

}  // namespace sympa_hip
