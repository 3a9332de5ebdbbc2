// gfx950 kernels + C-ABI for the optimiser-side manifold operations over the embedding table
// (SURVEY 8f-2): egrad2rgrad, projx, and the fused RiemannianSGD step.  One table row per lane; these run
// once per optimiser step over N rows (N = 5 041 .. 100 000) and are bandwidth-trivial next to the
// distance kernels, so rows are loaded directly.
#include <cstdint>
#include <cstdlib>

#include "siegel_table_kernel.hpp"

namespace {
using namespace sympa_hip;

// OP 0: out = projx(z).   OP 1: table <- retr(table, -lr * egrad2rgrad(table, grad + wd * table)) in place.
// sum of squares of `count` doubles, accumulated into acc[0] (the total gradient norm of clip_grad_norm_)
// Four independent 16-byte loads in flight per lane and four partial sums: the one-load-at-a-time form (a dependent load -> fma chain,
// 22 trips per lane over the 46.6 MB gradient of configs[3]) ran at 0.8 TB/s, 58 us of a 0.82 ms training step; this one at the
// copy rate.  (The order of summation is not fixed either way: the blocks add their sums with an atomic.)  A first version of
// this rewrite with 2 048 blocks and one atomic per wave was SLOWER (105 us): see the block-level sum at the end.
__global__ __launch_bounds__(BLOCK) void sqnorm_kernel(const double* __restrict__ x, int64_t count, double* acc) {
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x, nth = (int64_t)gridDim.x * BLOCK;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        const v2d* __restrict__ x2 = reinterpret_cast<const v2d*>(x);
        const int64_t n2 = count >> 1;
        int64_t i = tid;
        for (; i + 3 * nth < n2; i += 4 * nth) {
            const v2d a = x2[i], b = x2[i + nth], c = x2[i + 2 * nth], d = x2[i + 3 * nth];
            s0 = fma(a.x, a.x, fma(a.y, a.y, s0));
            s1 = fma(b.x, b.x, fma(b.y, b.y, s1));
            s2 = fma(c.x, c.x, fma(c.y, c.y, s2));
            s3 = fma(d.x, d.x, fma(d.y, d.y, s3));
        }
        for (; i < n2; i += nth) {
            const v2d a = x2[i];
            s0 = fma(a.x, a.x, fma(a.y, a.y, s0));
        }
        if (tid == 0 && (count & 1)) s1 = fma(x[count - 1], x[count - 1], s1);
    } else {
        for (int64_t i = tid; i < count; i += nth) s0 = fma(x[i], x[i], s0);
    }
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    // ONE atomic per block: the 4 096 per-wave atomics of the old form all hit the same word and serialise there (~12 ns each:
    // that, not the 46.6 MB read, was most of the 58 us)
    __shared__ double part[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) t += part[w];
        if (t != 0.0) atomicAdd(acc, t);
    }
}

// p <- p - lr * (coef * grad + wd * p), coef = min(1, max_norm / (sqrt(total_sqnorm) + 1e-6)): the plain SGD step of the
// parameters that live on no manifold (the model's scale), with the same folded gradient clip as the table step
__global__ __launch_bounds__(BLOCK) void sgd_step_kernel(double* __restrict__ p, const double* __restrict__ g, int64_t count, double lr,
                                                         double wd, const double* __restrict__ clip, double max_norm) {
    const double coef = (clip != nullptr) ? fmin(1.0, max_norm / (sqrt(clip[0]) + 1e-6)) : 1.0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (int64_t)gridDim.x * BLOCK)
        p[i] = fma(-lr, fma(wd, p[i], coef * g[i]), p[i]);
}

// grad[idx[r]] += alpha * rows[r] for `count` rows of `rowd` doubles: consecutive lanes take consecutive doubles of a
// row, so one atomic wave-instruction covers 512 contiguous bytes of ONE gradient row (the shape the fp64 atomics run
// at rate with, profiles/r01_atomic_scope.txt); rows with an index outside [0, num_rows) are skipped and flagged.
__global__ __launch_bounds__(BLOCK) void scatter_add_rows_kernel(const double* __restrict__ rows, const int64_t* __restrict__ idx,
                                                                 int64_t idx_stride, int64_t count, int rowd, int64_t num_rows,
                                                                 double alpha, double* __restrict__ grad, int32_t* status) {
    const int64_t total = count * rowd;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t r = t / rowd;
        const int e = (int)(t - r * rowd);
        const int64_t row = idx[r * idx_stride];
        if (row < 0 || row >= num_rows) {
            if (e == 0 && status != nullptr) { atomicOr(&status[0], sympa::ST_BAD_INDEX); atomicAdd(&status[1], 1); }
            continue;
        }
        const double v = rows[t] * alpha;
        if (v != 0.0) atomicAdd(grad + row * rowd + e, v);
    }
}

// Deterministic counterpart of the atomic scatter: grad[r] (=|+=) alpha * sum of the per-pair gradient rows that belong to
// table row r, added IN THE ORDER of a precomputed list (order[rowptr[r] .. rowptr[r + 1]) = the slots of row r in `rows`,
// sorted once per epoch on the host side of the step).  One thread per (table row, element): consecutive lanes read
// consecutive doubles of a slot row.  No atomics, every element of grad is written by exactly one thread, so two runs give
// the same bits; untouched rows are written as 0 (accumulate = 0): no separate zeroing pass either.
// The last block (when wave_partials is given) adds the per-wave sums of the loss, of d loss / d scale and of
// d loss / d w_k that the backward kernel left (siegel_bwd_kernel.hpp) in a fixed order as well.
__global__ __launch_bounds__(BLOCK) void segment_sum_rows_kernel(const double* __restrict__ rows, const int32_t* __restrict__ order,
                                                                 const int32_t* __restrict__ rowptr, int64_t num_rows, int rowd,
                                                                 int64_t order_stride, const int64_t* __restrict__ counter,
                                                                 double alpha, int accumulate, double* __restrict__ grad,
                                                                 const double* __restrict__ wave_partials, int64_t num_waves,
                                                                 int quantities, int partial_stride, double* loss, double* gscale,
                                                                 double* gw, unsigned row_blocks, double* sq_partials) {
    __shared__ double red[BLOCK];
    if (blockIdx.x >= row_blocks) {
        double extra_sq = 0.0;
        // fixed-order sums of the per-wave partials: thread t takes waves t, t + 256, ...; then a tree over the block
        for (int k = 0; k < quantities; ++k) {
            double s = 0.0;
            for (int64_t w = threadIdx.x; w < num_waves; w += BLOCK) s += wave_partials[w * partial_stride + k];
            red[threadIdx.x] = s;
            __syncthreads();
            for (int off = BLOCK / 2; off > 0; off >>= 1) {
                if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                double* dst = (k == 0) ? loss : (k == 1 ? gscale : (gw != nullptr ? gw + (k - 2) : nullptr));
                if (dst != nullptr) {
                    const double v = dst[0] + red[0];
                    dst[0] = v;
                    if (k >= 1) extra_sq = fma(v, v, extra_sq);      // the scale's and the weights' gradients enter the clip norm
                }
            }
            __syncthreads();
        }
        if (sq_partials != nullptr && threadIdx.x == 0) sq_partials[row_blocks] = extra_sq;
        return;
    }
    const int64_t c = counter != nullptr ? counter[0] : 0;
    order += c * order_stride;
    rowptr += c * (num_rows + 1);
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = t < num_rows * rowd;      // no early return: the block's tree below needs every thread at its barriers
    double v = 0.0;
    if (in_range) {
        const int64_t r = t / rowd;
        const int e = (int)(t - r * rowd);
        const int p0 = rowptr[r], p1 = rowptr[r + 1];
        double s = 0.0;
        int p = p0;
        for (; p + 8 <= p1; p += 8) {       // eight loads in flight, added in list order
            double x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = rows[(int64_t)order[p + k] * rowd + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += x[k];
        }
        for (; p + 4 <= p1; p += 4) {
            const double v0 = rows[(int64_t)order[p] * rowd + e], v1 = rows[(int64_t)order[p + 1] * rowd + e];
            const double v2 = rows[(int64_t)order[p + 2] * rowd + e], v3 = rows[(int64_t)order[p + 3] * rowd + e];
            s = ((s + v0) + v1) + v2 + v3;
        }
        for (; p < p1; ++p) s += rows[(int64_t)order[p] * rowd + e];
        s *= alpha;
        v = accumulate ? grad[t] + s : s;
        grad[t] = v;
    }
    if (sq_partials != nullptr) {
        // squared norm of the finished gradient, one partial per block in a fixed tree: the optimiser kernel adds the
        // partials in index order and needs neither its own pass over the gradient nor a grid barrier
        // (a fixed tree: the xor tree of every wave, then the waves in index order -- one barrier instead of eight)
        double q = v * v;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = red[0];
            for (int w = 1; w < BLOCK / 64; ++w) t += red[w];
            sq_partials[blockIdx.x] = t;
        }
    }
}

int dispatch_table(int op, int n, int model, double* z, const double* g, double* out, int64_t b, double lr, double wd,
                   double eps, int32_t* projected, int32_t* status, void* stream, const double* clip = nullptr,
                   double max_norm = 0.0, int32_t* outside = nullptr) {
    if (b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative row count");
    if (b == 0) return 0;
    if (z == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (b > (int64_t)0x7fffffff * BLOCK) return fail(SYMPA_ERR_BAD_ARG, "too many rows for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const bool generic = std::getenv("SYMPA_TABLE_GENERIC") != nullptr;      // one-row-per-lane kernels for A/B
    // projx / the RSGD step at dims >= 7 need the caller's scratch word `outside`; without it they run the one-row-per-lane
    // kernels below (exact, every row through the eigenvalue clamp, slower)
    const bool needs_word = (op == 0 || op == 1);
    if (n >= 7 && n <= SYMPA_MAX_DIMS_GENERIC && !generic && !instance_fallback(SYMPA_FAMILY_SIEGEL_TABLE, model, n) &&
        !(needs_word && outside == nullptr)) {
        // Eight (n = 7, 8) or sixteen (n = 9..16) lanes per row (siegel_coop_table.hpp).  egrad2rgrad and the tangent norm
        // are complete there.  The RSGD step and projx symmetrise, test "inside the eps-interior" by a Cholesky
        // factorisation and count the rows that fail in a device word; the exact one-row-per-lane projx (the reference's
        // eigenvalue clamp, which leaves inside rows untouched) then runs gated on that word -- it returns at once in the
        // usual case of zero.  No host synchronisation, no allocation, capturable in the training hipGraph.  The word is
        // the CALLER's (one per call in flight): the library keeps no state of its own, so the entry is re-entrant across
        // streams, devices and threads.
        const bool gated = needs_word;
        if (gated && hipMemsetAsync(outside, 0, sizeof(int), s) != hipSuccess) return fail(SYMPA_ERR_BAD_ARG, "memset failed");
        int* word = gated ? reinterpret_cast<int*>(outside) : (op == 3 ? reinterpret_cast<int*>(status) : nullptr);
        const bool up = model == SYMPA_MODEL_UPPER;
        int rc;
        if (n <= 8) rc = up ? launch_table_half_upper(op, n, z, g, out, b, lr, wd, eps, clip, max_norm, word, s)
                            : launch_table_half_bounded(op, n, z, g, out, b, lr, wd, eps, clip, max_norm, word, s);
        else rc = up ? launch_table_coop_upper(op, n, z, g, out, b, lr, wd, eps, clip, max_norm, word, s)
                     : launch_table_coop_bounded(op, n, z, g, out, b, lr, wd, eps, clip, max_norm, word, s);
        if (rc != 0 || !gated) return rc;
        double* target = (op == 0) ? out : z;        // projx wrote sym(z) to out; the step updated z in place
        if (n == 7) return launch_table<7>(0, model, target, nullptr, target, b, 0.0, 0.0, eps, projected, status, s, nullptr, 0.0, outside);
        if (n == 8) return launch_table<8>(0, model, target, nullptr, target, b, 0.0, 0.0, eps, projected, status, s, nullptr, 0.0, outside);
        return launch_table_rolled(0, n, model, target, nullptr, target, b, 0.0, 0.0, eps, projected, status, s, nullptr, 0.0, outside);
    }
    switch (n) {
        case 1: return launch_table<1>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        case 2: return launch_table<2>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        case 3: return launch_table<3>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        case 4: return launch_table<4>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        case 5: return launch_table<5>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        case 6: return launch_table<6>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        case 7: return launch_table<7>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        case 8: return launch_table<8>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
        default: break;
    }
    if (n > 8 && n <= SYMPA_MAX_DIMS_GENERIC)
        return launch_table_rolled(op, n, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm);
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "dims outside [1, SYMPA_MAX_DIMS_GENERIC]");
}

}  // namespace

extern "C" {

int sympa_egrad2rgrad(const double* z, const double* u, int64_t b, int n, int model, double* out, void* stream) {
    if (b > 0 && (u == nullptr || out == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    return dispatch_table(2, n, model, const_cast<double*>(z), u, out, b, 0.0, 0.0, 0.0, nullptr, nullptr, stream);
}

int sympa_tangent_sqnorm(const double* z, const double* u, int64_t b, int n, int model, double* out, int32_t* status,
                         void* stream) {
    if (b > 0 && (u == nullptr || out == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    return dispatch_table(3, n, model, const_cast<double*>(z), u, out, b, 0.0, 0.0, 0.0, nullptr, status, stream);
}

int sympa_projx(const double* z, int64_t b, int n, int model, double eps, double* out, int32_t* projected_count,
                int32_t* status, int32_t* outside_word, void* stream) {
    if (b > 0 && out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (!(eps > 0.0)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    return dispatch_table(0, n, model, const_cast<double*>(z), nullptr, out, b, 0.0, 0.0, eps, projected_count, status,
                          stream, nullptr, 0.0, outside_word);
}

int sympa_rsgd_step(double* table, const double* grad, int64_t num_rows, int n, int model, double lr,
                    double weight_decay, double eps, int32_t* projected_count, int32_t* status, int32_t* outside_word,
                    void* stream) {
    if (num_rows > 0 && grad == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null gradient");
    if (!(eps > 0.0)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    return dispatch_table(1, n, model, table, grad, nullptr, num_rows, lr, weight_decay, eps, projected_count, status,
                          stream, nullptr, 0.0, outside_word);
}

int sympa_radam_step(double* table, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t num_rows, int n, int model,
                     double lr, double beta1, double beta2, double eps_adam, double weight_decay, const double* bias_pows,
                     double eps, int32_t* projected_count, int32_t* status, void* stream) {
    if (num_rows < 0) return fail(SYMPA_ERR_BAD_ARG, "negative row count");
    if (num_rows == 0) return 0;
    if (table == nullptr || grad == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || bias_pows == nullptr)
        return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (!(eps > 0.0) || !(eps_adam >= 0.0)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0, the Adam epsilon >= 0");
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0)) return fail(SYMPA_ERR_BAD_ARG, "betas must lie in [0, 1)");
    if (num_rows > (int64_t)0x7fffffff * BLOCK) return fail(SYMPA_ERR_BAD_ARG, "too many rows for one launch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define SYMPA_RADAM_CASE(NN) case NN: return launch_radam<NN>(model, table, grad, exp_avg, exp_avg_sq, num_rows, lr, beta1, beta2, \
                                                              eps_adam, weight_decay, bias_pows, eps, projected_count, status, s);
    switch (n) {
        SYMPA_RADAM_CASE(1) SYMPA_RADAM_CASE(2) SYMPA_RADAM_CASE(3) SYMPA_RADAM_CASE(4) SYMPA_RADAM_CASE(5) SYMPA_RADAM_CASE(6)
        default: break;
    }
#undef SYMPA_RADAM_CASE
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "sympa_radam_step: dims 1..6 (larger tables: the separate row kernels)");
}

int sympa_sqnorm_accum(const double* x, int64_t count, double* acc, void* stream) {
    if (count < 0) return fail(SYMPA_ERR_BAD_ARG, "negative count");
    if (count == 0) return 0;
    if (x == nullptr || acc == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    const int64_t want = (count / 8 + BLOCK - 1) / BLOCK;          // one trip of the four-loads loop per lane, up to 2 blocks per CU
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want < 512 ? want : 512));
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream), x, count, acc);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

int sympa_sgd_step_clipped(double* p, const double* grad, int64_t count, double lr, double weight_decay,
                           const double* total_sqnorm, double max_norm, void* stream) {
    if (count < 0) return fail(SYMPA_ERR_BAD_ARG, "negative count");
    if (count == 0) return 0;
    if (p == nullptr || grad == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (total_sqnorm != nullptr && !(max_norm > 0.0)) return fail(SYMPA_ERR_BAD_ARG, "max_norm must be > 0");
    const int64_t want = (count + BLOCK - 1) / BLOCK;
    const unsigned grid = (unsigned)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(sgd_step_kernel, dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream), p, grad, count, lr,
                       weight_decay, total_sqnorm, max_norm);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

int sympa_scatter_add_rows(const double* rows, const int64_t* idx, int64_t idx_stride, int64_t count, int n,
                           int64_t num_rows, double alpha, double* grad_table, int32_t* status, void* stream) {
    if (n < 1 || n > SYMPA_MAX_DIMS_GENERIC) return fail(SYMPA_ERR_BAD_ARG, "bad dims");
    return sympa_scatter_add_flat_rows(rows, idx, idx_stride, count, 2 * n * n, num_rows, alpha, grad_table, status, stream);
}

int sympa_scatter_add_flat_rows(const double* rows, const int64_t* idx, int64_t idx_stride, int64_t count, int row_doubles,
                                int64_t num_rows, double alpha, double* grad_table, int32_t* status, void* stream) {
    if (count < 0 || row_doubles < 1 || row_doubles > 2 * SYMPA_MAX_DIMS_GENERIC * SYMPA_MAX_DIMS_GENERIC)
        return fail(SYMPA_ERR_BAD_ARG, "bad row count / row length");
    if (count == 0) return 0;
    if (rows == nullptr || idx == nullptr || grad_table == nullptr || num_rows <= 0)
        return fail(SYMPA_ERR_BAD_ARG, "null buffer / empty table");
    const int rowd = row_doubles;
    const int64_t want = (count * rowd + BLOCK - 1) / BLOCK;
    const unsigned grid = (unsigned)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream), rows, idx,
                       idx_stride, count, rowd, num_rows, alpha, grad_table, status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

int64_t sympa_segment_sum_partials(int64_t num_rows, int row_doubles) {
    if (num_rows <= 0 || row_doubles < 1) return 0;
    return (num_rows * row_doubles + BLOCK - 1) / BLOCK + 1;
}

int sympa_segment_sum_rows(const double* rows, const int32_t* order, const int32_t* rowptr, int64_t num_rows, int row_doubles,
                           int64_t order_stride, const int64_t* step_counter, double alpha, int accumulate, double* grad_table,
                           const double* wave_partials, int64_t num_waves, int partial_stride, int num_weights, double* loss,
                           double* grad_scale, double* grad_w, double* sq_partials, void* stream) {
    if (num_rows <= 0 || row_doubles < 1 || row_doubles > 2 * SYMPA_MAX_DIMS_GENERIC * SYMPA_MAX_DIMS_GENERIC)
        return fail(SYMPA_ERR_BAD_ARG, "bad table shape");
    if (rows == nullptr || order == nullptr || rowptr == nullptr || grad_table == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (num_rows * row_doubles > (int64_t)0x7fffffff * BLOCK) return fail(SYMPA_ERR_BAD_ARG, "table too large for one launch");
    if (wave_partials != nullptr && (num_waves < 0 || num_weights < 0 || partial_stride < 2 + num_weights || loss == nullptr ||
                                     (num_weights > 0 && grad_w == nullptr)))
        return fail(SYMPA_ERR_BAD_ARG, "bad partial-sum arguments");
    if (sq_partials != nullptr && wave_partials == nullptr)
        return fail(SYMPA_ERR_BAD_ARG, "sq_partials needs wave_partials (the scalar gradients enter the clip norm)");
    const unsigned row_blocks = (unsigned)((num_rows * row_doubles + BLOCK - 1) / BLOCK);
    const unsigned grid = row_blocks + (wave_partials != nullptr ? 1u : 0u);
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3(grid), dim3(BLOCK), 0, reinterpret_cast<hipStream_t>(stream), rows, order,
                       rowptr, num_rows, row_doubles, order_stride, step_counter, alpha, accumulate, grad_table, wave_partials,
                       num_waves, 2 + num_weights, partial_stride, loss, grad_scale, grad_w, row_blocks, sq_partials);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

}  // extern "C"

namespace {
struct AdamSide {       // the RiemannianAdam state sympa_radam_step_fused adds to the fused step; null: RiemannianSGD
    double* exp_avg;
    double* exp_avg_sq;
    double* bias_pows;
    double* const* extra_exp_avg;
    double* const* extra_exp_avg_sq;
    double* const* extra_bias_pows;
    double beta1, beta2, eps_adam;
};

int fused_step_impl(double* table, double* grad, int64_t num_rows, int n, int model, double lr, double weight_decay,
                    double eps, double max_norm, int zero_grads, double* const* extra_param, double* const* extra_grad,
                    const int* extra_count, const double* extra_lr, const double* extra_weight_decay, int num_extra,
                    void* workspace, int64_t workspace_bytes, const double* sq_partials, int num_sq_partials,
                    int64_t* step_counter, int32_t* projected_count, int32_t* status, void* stream, const AdamSide* adam) {
    if (num_rows <= 0 || table == nullptr || grad == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer / empty table");
    if (adam != nullptr) {
        if (adam->exp_avg == nullptr || adam->exp_avg_sq == nullptr || adam->bias_pows == nullptr)
            return fail(SYMPA_ERR_BAD_ARG, "null Adam state");
        if (!(adam->beta1 >= 0.0 && adam->beta1 < 1.0 && adam->beta2 >= 0.0 && adam->beta2 < 1.0) || !(adam->eps_adam >= 0.0))
            return fail(SYMPA_ERR_BAD_ARG, "betas must lie in [0, 1), the Adam epsilon must be >= 0");
        if (num_extra > 0 && (adam->extra_exp_avg == nullptr || adam->extra_exp_avg_sq == nullptr || adam->extra_bias_pows == nullptr))
            return fail(SYMPA_ERR_BAD_ARG, "null Adam state of the plain parameters");
    }
    if (sq_partials != nullptr && num_sq_partials < 1) return fail(SYMPA_ERR_BAD_ARG, "empty partial list");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (!(eps > 0.0)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    if (num_extra < 0 || num_extra > FUSED_MAX_EXTRA) return fail(SYMPA_ERR_BAD_ARG, "at most 2 plain parameters");
    if (n < 1 || n > 6) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "fused step: dims 1..6 (one row per lane)");
    // the grid barrier needs every block resident at once: one block per CU is always possible.  One-wave blocks while
    // they fit (a wave per CU instead of four on a quarter of the CUs), 256-thread blocks above
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return fail(SYMPA_ERR_BAD_ARG, "no device");
    const int block = (num_rows + 63) / 64 <= cus ? 64 : BLOCK;
    const int64_t grid = (num_rows + block - 1) / block;
    if (grid > cus) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "fused step: more row blocks than CUs (use sympa_rsgd_step_clipped)");
    const int64_t need = sympa_rsgd_step_fused_workspace_bytes(num_rows);
    if (workspace == nullptr || workspace_bytes < need) return fail(SYMPA_ERR_BAD_ARG, "workspace too small");
    FusedStepArgs a;
    std::memset(&a, 0, sizeof(a));
    a.table = table; a.grad = grad; a.rows = num_rows;
    a.lr = lr; a.wd = weight_decay; a.eps = eps; a.max_norm = max_norm;
    a.sync = reinterpret_cast<unsigned*>(workspace);
    a.partial = reinterpret_cast<double*>(workspace) + 1;
    for (int k = 0; k < num_extra; ++k) {
        if (extra_param == nullptr || extra_grad == nullptr || extra_count == nullptr || extra_lr == nullptr ||
            extra_weight_decay == nullptr || extra_param[k] == nullptr || extra_grad[k] == nullptr)
            return fail(SYMPA_ERR_BAD_ARG, "null plain parameter");
        if (extra_count[k] < 1 || extra_count[k] > 64) return fail(SYMPA_ERR_BAD_ARG, "plain parameters of 1..64 elements");
        a.xp[k] = extra_param[k]; a.xg[k] = extra_grad[k]; a.xn[k] = extra_count[k];
        a.xlr[k] = extra_lr[k]; a.xwd[k] = extra_weight_decay[k];
    }
    a.counter = step_counter; a.projected = projected_count; a.status = status; a.zero_grads = zero_grads;
    a.sq_in = sq_partials; a.sq_in_count = num_sq_partials;
    if (adam != nullptr) {
        a.am = adam->exp_avg; a.av = adam->exp_avg_sq; a.apows = adam->bias_pows;
        a.b1 = adam->beta1; a.b2 = adam->beta2; a.aeps = adam->eps_adam;
        for (int k = 0; k < num_extra; ++k) {
            if (adam->extra_exp_avg[k] == nullptr || adam->extra_exp_avg_sq[k] == nullptr || adam->extra_bias_pows[k] == nullptr)
                return fail(SYMPA_ERR_BAD_ARG, "null Adam state of a plain parameter");
            a.xam[k] = adam->extra_exp_avg[k]; a.xav[k] = adam->extra_exp_avg_sq[k]; a.xapows[k] = adam->extra_bias_pows[k];
        }
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (n) {
        case 1: return launch_fused_step<1>(a, model, block, s);
        case 2: return launch_fused_step<2>(a, model, block, s);
        case 3: return launch_fused_step<3>(a, model, block, s);
        case 4: return launch_fused_step<4>(a, model, block, s);
        case 5: return launch_fused_step<5>(a, model, block, s);
        default: return launch_fused_step<6>(a, model, block, s);
    }
}
}  // namespace

extern "C" {

int sympa_rsgd_step_fused(double* table, double* grad, int64_t num_rows, int n, int model, double lr, double weight_decay,
                          double eps, double max_norm, int zero_grads, double* const* extra_param, double* const* extra_grad,
                          const int* extra_count, const double* extra_lr, const double* extra_weight_decay, int num_extra,
                          void* workspace, int64_t workspace_bytes, const double* sq_partials, int num_sq_partials,
                          int64_t* step_counter, int32_t* projected_count, int32_t* status, void* stream) {
    return fused_step_impl(table, grad, num_rows, n, model, lr, weight_decay, eps, max_norm, zero_grads, extra_param, extra_grad,
                           extra_count, extra_lr, extra_weight_decay, num_extra, workspace, workspace_bytes, sq_partials,
                           num_sq_partials, step_counter, projected_count, status, stream, nullptr);
}

int sympa_radam_step_fused(double* table, double* grad, double* exp_avg, double* exp_avg_sq, double* bias_pows, int64_t num_rows,
                           int n, int model, double lr, double beta1, double beta2, double eps_adam, double weight_decay,
                           double eps, double max_norm, int zero_grads, double* const* extra_param, double* const* extra_grad,
                           double* const* extra_exp_avg, double* const* extra_exp_avg_sq, double* const* extra_bias_pows,
                           const int* extra_count, const double* extra_lr, const double* extra_weight_decay, int num_extra,
                           void* workspace, int64_t workspace_bytes, const double* sq_partials, int num_sq_partials,
                           int64_t* step_counter, int32_t* projected_count, int32_t* status, void* stream) {
    const AdamSide adam{exp_avg, exp_avg_sq, bias_pows, extra_exp_avg, extra_exp_avg_sq, extra_bias_pows, beta1, beta2, eps_adam};
    return fused_step_impl(table, grad, num_rows, n, model, lr, weight_decay, eps, max_norm, zero_grads, extra_param, extra_grad,
                           extra_count, extra_lr, extra_weight_decay, num_extra, workspace, workspace_bytes, sq_partials,
                           num_sq_partials, step_counter, projected_count, status, stream, &adam);
}

int64_t sympa_rsgd_step_fused_workspace_bytes(int64_t num_rows) {
    const int64_t grid = (num_rows + 63) / 64;                 // the largest grid any block size gives
    return 8 * (grid + 2);          // one 8-byte slot for the two barrier words, grid + 1 partial sums
}

int sympa_rsgd_step_clipped(double* table, const double* grad, int64_t num_rows, int n, int model, double lr,
                            double weight_decay, double eps, const double* total_sqnorm, double max_norm,
                            int32_t* projected_count, int32_t* status, int32_t* outside_word, void* stream) {
    if (num_rows > 0 && (grad == nullptr || total_sqnorm == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (!(eps > 0.0) || !(max_norm > 0.0)) return fail(SYMPA_ERR_BAD_ARG, "eps and max_norm must be > 0");
    return dispatch_table(1, n, model, table, grad, nullptr, num_rows, lr, weight_decay, eps, projected_count, status,
                          stream, total_sqnorm, max_norm, outside_word);
}

}  // extern "C"
