// Kernel templates of the optimiser-side table operations (siegel_table_math.hpp), one table row per lane; instantiated
// by siegel_table.hip (n <= 8, fully unrolled, registers) and siegel_table_rolled.hip (n = 9..16, SYMPA_UNROLL = nounroll:
// rolled loops over per-lane scratch arrays -- functional, not tuned, like siegel_bwd_rolled.hip).
#pragma once
#include "siegel_common.hpp"
#include "siegel_table_math.hpp"

namespace sympa_hip {

template <int N, int MODEL>
__global__ __launch_bounds__(BLOCK) void egrad2rgrad_kernel(const double* z, const double* u, double* out, int64_t b) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b) return;
    constexpr int64_t ROW = 2 * N * N;
    sympa::CMat<N> a, g, r;
    sympa::load_full<N>(z + i * ROW, a);
    sympa::load_full<N>(u + i * ROW, g);
    sympa::egrad2rgrad<N, MODEL>(a, g, r);
    sympa::store_full<N>(out + i * ROW, r);
}

template <int N, int MODEL>
__global__ __launch_bounds__(BLOCK) void tangent_sqnorm_kernel(const double* z, const double* u, double* out, int64_t b,
                                                               int32_t* status) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b) return;
    constexpr int64_t ROW = 2 * N * N;
    sympa::CMat<N> a, g;
    sympa::load_point<N>(z + i * ROW, a);          // the point is symmetric: upper triangle
    sympa::load_full<N>(u + i * ROW, g);
    int st = 0;
    out[i] = sympa::tangent_sqnorm<N, MODEL>(a, g, st);
    if (st != 0 && status != nullptr) { atomicOr(&status[0], st); atomicAdd(&status[1], 1); }
}

// clip: device pointer to the squared total gradient norm, or null.  torch.nn.utils.clip_grad_norm_ (runner.py:115)
// scales every gradient by min(1, max_norm / (total_norm + 1e-6)); here the factor is applied to the row's gradient
// as it is loaded, so the clip costs no pass of its own.
template <int N, int MODEL, int OP>
__global__ __launch_bounds__(BLOCK) void table_update_kernel(double* z, const double* grad, double* out, int64_t b,
                                                             double lr, double wd, double eps, int32_t* projected,
                                                             int32_t* status, const double* clip, double max_norm,
                                                             const int* gate) {
    // gate: device word written by the sixteen-lanes step kernel (siegel_coop_table.hpp) = rows that left the interior;
    // zero (the usual case) -> nothing to project
    if (gate != nullptr && *gate == 0) return;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < b;
    const int64_t ii = live ? i : b - 1;      // tail lanes recompute the last row (the Jacobi loops ballot)
    constexpr int64_t ROW = 2 * N * N;
    sympa::CMat<N> a;
    sympa::load_full<N>(z + ii * ROW, a);
    int st = 0;
    bool moved;
    if (OP == 0) {
        moved = sympa::projx<N, MODEL>(a, eps, st);
    } else {
        sympa::CMat<N> g;
        sympa::load_full<N>(grad + ii * ROW, g);
        if (clip != nullptr) {
            const double coef = fmin(1.0, max_norm / (sqrt(clip[0]) + 1e-6));
SYMPA_UNROLL
            for (int r = 0; r < N; ++r)
SYMPA_UNROLL
                for (int c = 0; c < N; ++c) { g.re[r][c] *= coef; g.im[r][c] *= coef; }
        }
        moved = sympa::rsgd_row<N, MODEL>(a, g, lr, wd, eps, st);
    }
    if (live) sympa::store_full<N>((OP == 0 ? out : z) + i * ROW, a);
    const unsigned long long m = __ballot(live && moved);
    if (projected != nullptr && m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(projected, (int)__popcll(m));
    if (status != nullptr) {
        const unsigned long long f = __ballot(live && st != 0);
        if (f != 0ull) {
            if (live && st != 0) atomicOr(&status[0], st);
            if ((threadIdx.x & 63) == 0) atomicAdd(&status[1], (int)__popcll(f));
        }
    }
}

// One RiemannianAdam step over the rows of the table, one row per lane (geoopt.optim.RiemannianAdam as train.py:69-70 builds
// it; restated, geoopt is absent from the reference tree):
//     g <- egrad2rgrad(x, grad + wd x);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) inner(x, g, g)   (one v per row);
//     x <- projx(x - lr (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps_adam))
// pows: device words {b1^t, b2^t} (already advanced to this step by the caller): no host state in the kernel's arguments
// changes from step to step, so the launch can sit in a replayed hipGraph.
// The row update itself.  In: x = the point, g = the (clipped) Euclidean gradient row.  Out: x = the new point, g = the new
// first moment, returns the new second moment through vn; `moved` = projx changed the point.
template <int N, int MODEL>
__device__ __forceinline__ bool radam_row_update(sympa::CMat<N>& x, sympa::CMat<N>& g, const double* __restrict__ m_row,
                                                 const double v_old, double& vn, double lr, double b1, double b2,
                                                 double eps_adam, double wd, double pow1, double pow2, double eps, int& st) {
    sympa::CMat<N> r;
    if (wd != 0.0) {
SYMPA_UNROLL
        for (int a = 0; a < N; ++a)
SYMPA_UNROLL
            for (int c = 0; c < N; ++c) {
                g.re[a][c] = sympa::d_fma(wd, x.re[a][c], g.re[a][c]);
                g.im[a][c] = sympa::d_fma(wd, x.im[a][c], g.im[a][c]);
            }
    }
    sympa::egrad2rgrad<N, MODEL>(x, g, r);
    const double s = sympa::tangent_sqnorm<N, MODEL>(x, r, st);
    vn = sympa::d_fma(b2, v_old, (1.0 - b2) * s);
    const double bc1 = 1.0 - pow1, bc2 = 1.0 - pow2;
    const double step = lr / (bc1 * (sqrt(vn / bc2) + eps_adam));
    // m <- b1 m + (1 - b1) r (kept in g), x <- x - step m
    sympa::load_full<N>(m_row, g);
SYMPA_UNROLL
    for (int a = 0; a < N; ++a)
SYMPA_UNROLL
        for (int c = 0; c < N; ++c) {
            g.re[a][c] = sympa::d_fma(b1, g.re[a][c], (1.0 - b1) * r.re[a][c]);
            g.im[a][c] = sympa::d_fma(b1, g.im[a][c], (1.0 - b1) * r.im[a][c]);
            x.re[a][c] = sympa::d_fma(-step, g.re[a][c], x.re[a][c]);
            x.im[a][c] = sympa::d_fma(-step, g.im[a][c], x.im[a][c]);
        }
    return sympa::projx<N, MODEL>(x, eps, st);
}

template <int N, int MODEL>
__global__ __launch_bounds__(BLOCK) void radam_row_kernel(double* z, const double* grad, double* m, double* v, int64_t b,
                                                          double lr, double b1, double b2, double eps_adam, double wd,
                                                          const double* __restrict__ pows, double eps, int32_t* projected,
                                                          int32_t* status) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < b;
    const int64_t ii = live ? i : b - 1;      // tail lanes recompute the last row (the Jacobi loops ballot)
    constexpr int64_t ROW = 2 * N * N;
    sympa::CMat<N> x, g;
    sympa::load_full<N>(z + ii * ROW, x);
    sympa::load_full<N>(grad + ii * ROW, g);
    int st = 0;
    double vn;
    const bool moved = radam_row_update<N, MODEL>(x, g, m + ii * ROW, v[ii], vn, lr, b1, b2, eps_adam, wd, pows[0], pows[1], eps, st);
    if (live) {
        sympa::store_full<N>(m + i * ROW, g);
        v[i] = vn;
        sympa::store_full<N>(z + i * ROW, x);
    }
    const unsigned long long mm = __ballot(live && moved);
    if (projected != nullptr && mm != 0ull && (threadIdx.x & 63) == 0) atomicAdd(projected, (int)__popcll(mm));
    if (status != nullptr) {
        const unsigned long long f = __ballot(live && st != 0);
        if (f != 0ull) {
            if (live && st != 0) atomicOr(&status[0], st);
            if ((threadIdx.x & 63) == 0) atomicAdd(&status[1], (int)__popcll(f));
        }
    }
}

template <int N>
int launch_radam(int model, double* z, const double* grad, double* m, double* v, int64_t b, double lr, double b1, double b2,
                 double eps_adam, double wd, const double* pows, double eps, int32_t* projected, int32_t* status, hipStream_t s) {
    const unsigned grid = (unsigned)((b + BLOCK - 1) / BLOCK);
    if (model == SYMPA_MODEL_UPPER)
        hipLaunchKernelGGL((radam_row_kernel<N, sympa::MODEL_UPPER>), dim3(grid), dim3(BLOCK), 0, s, z, grad, m, v, b, lr, b1, b2,
                           eps_adam, wd, pows, eps, projected, status);
    else
        hipLaunchKernelGGL((radam_row_kernel<N, sympa::MODEL_BOUNDED>), dim3(grid), dim3(BLOCK), 0, s, z, grad, m, v, b, lr, b1, b2,
                           eps_adam, wd, pows, eps, projected, status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The optimiser side of one training step as ONE kernel (runner.py:113-118: clip_grad_norm_, optimizer.step, zero_grad):
//   phase 1  every block writes the sum of squares of its rows' gradients to partial[blockIdx]; block 0 adds the plain
//            parameters' (scale, wsum weights) to partial[gridDim];
//   barrier  all blocks are resident (the host refuses a grid larger than the number of CUs), one counter word;
//   phase 2  every block sums the partials IN INDEX ORDER (bitwise the same total in every block and in every run),
//            coef = min(1, max_norm / (sqrt(total) + 1e-6)), RiemannianSGD step of its rows, gradient rows zeroed for the
//            next step; block 0 steps the plain parameters (sympa_sgd_step_clipped's formula) and zeroes their gradients;
//   exit     the last block to finish resets the barrier words and increments the device step counter (the batch the
//            next replay of the training graph reads).
// Replaces five graph nodes (zero, two squared norms, the table step, the scale's step) by one.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FUSED_MAX_EXTRA = 2;
struct FusedStepArgs {
    double* table;
    double* grad;
    int64_t rows;
    double lr, wd, eps, max_norm;      // max_norm <= 0: no clip (then no barrier either)
    double* partial;                   // [gridDim + 1] workspace
    unsigned* sync;                    // [2], zero on entry, zero on exit
    double* xp[FUSED_MAX_EXTRA];       // plain parameters (<= 64 elements each), or null
    double* xg[FUSED_MAX_EXTRA];
    int xn[FUSED_MAX_EXTRA];
    double xlr[FUSED_MAX_EXTRA], xwd[FUSED_MAX_EXTRA];
    int64_t* counter;                  // device step counter (+= 1 on exit) or null
    int32_t* projected;
    int32_t* status;
    int zero_grads;
    const double* sq_in;               // squared-norm partials already computed by sympa_segment_sum_rows (deterministic
    int sq_in_count;                   // mode): phase 1 and the barrier are skipped, the partials are summed in index order
    // RiemannianAdam (fused_step_kernel<.., ADAM = true>): moments of the table and of the plain parameters, the device
    // words {b1^t, b2^t} of each (read at the start, advanced by the last block to finish), hyper-parameters
    double* am;                        // [rows, 2, n, n]
    double* av;                        // [rows]
    double* apows;                     // [2]
    double* xam[FUSED_MAX_EXTRA];
    double* xav[FUSED_MAX_EXTRA];
    double* xapows[FUSED_MAX_EXTRA];
    double b1, b2, aeps;
};

// FB = threads per block: 64 while the table has at most as many 64-row groups as the chip has CUs (one wave per CU: a
// lane-per-row load or store touches 64 cache lines per instruction, so four waves on one CU queue behind one texture
// addresser -- 5 041 rows at n = 4: 15.5 us with 20 blocks of 256, measured), 256 above that (the barrier needs grid <= CUs).
template <int N, int MODEL, int FB, bool ADAM = false>
__global__ __launch_bounds__(FB) void fused_step_kernel(const FusedStepArgs a) {
    constexpr int BLOCK = FB;
    __shared__ double red[BLOCK / 64];
    __shared__ double total_s;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < a.rows;
    const int64_t ii = live ? i : a.rows - 1;
    constexpr int64_t ROW = 2 * N * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    sympa::CMat<N> z, g;
    sympa::load_full<N>(a.table + ii * ROW, z);
    sympa::load_full<N>(a.grad + ii * ROW, g);
    int st = 0;
    double coef = 1.0;
    if (a.max_norm > 0.0 && a.sq_in != nullptr) {
        if (wave == 0) {
            // lane l adds partials l, l + 64, ... in index order (eight loads in flight: a few thousand partials are a chain of
            // dependent L2 round trips otherwise), then the xor tree -- the same bits in every block
            double t = 0.0;
            int k = lane;
            for (; k + 7 * 64 < a.sq_in_count; k += 8 * 64) {
                double x[8];
SYMPA_UNROLL
                for (int j = 0; j < 8; ++j) x[j] = a.sq_in[k + 64 * j];
SYMPA_UNROLL
                for (int j = 0; j < 8; ++j) t += x[j];
            }
            for (; k < a.sq_in_count; k += 64) t += a.sq_in[k];
SYMPA_UNROLL
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if (lane == 0) total_s = t;
        }
        __syncthreads();
        coef = fmin(1.0, a.max_norm / (sqrt(total_s) + 1e-6));
SYMPA_UNROLL
        for (int r = 0; r < N; ++r)
SYMPA_UNROLL
            for (int c = 0; c < N; ++c) { g.re[r][c] *= coef; g.im[r][c] *= coef; }
    } else if (a.max_norm > 0.0) {
        double s = 0.0;
SYMPA_UNROLL
        for (int r = 0; r < N; ++r)
SYMPA_UNROLL
            for (int c = 0; c < N; ++c) s = fma(g.re[r][c], g.re[r][c], fma(g.im[r][c], g.im[r][c], s));
        s = live ? s : 0.0;
SYMPA_UNROLL
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave] = s;
        double xs = 0.0;
        if (blockIdx.x == 0 && wave == 0) {
SYMPA_UNROLL
            for (int k = 0; k < FUSED_MAX_EXTRA; ++k)
                if (a.xg[k] != nullptr && lane < a.xn[k]) { const double v = a.xg[k][lane]; xs = fma(v, v, xs); }
SYMPA_UNROLL
            for (int off = 32; off > 0; off >>= 1) xs += __shfl_xor(xs, off);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
SYMPA_UNROLL
            for (int w = 0; w < BLOCK / 64; ++w) t += red[w];
            __hip_atomic_store(a.partial + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) __hip_atomic_store(a.partial + gridDim.x, xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // grid barrier: release my partial, count myself, wait for everybody (bounded: a stranded block must not hang
            // the device -- it raises the status word instead and the step is garbage)
            __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(a.sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { st |= sympa::ST_NO_CONVERGENCE; break; }
            }
        }
        __syncthreads();
        if (wave == 0) {
            // fixed order: lane l sums partial[l], partial[l + 64], ...; then the xor tree -- the same bits in every block
            double t = 0.0;
            for (unsigned k = lane; k <= gridDim.x; k += 64)
                t += __hip_atomic_load(a.partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
SYMPA_UNROLL
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if (lane == 0) total_s = t;
        }
        __syncthreads();
        coef = fmin(1.0, a.max_norm / (sqrt(total_s) + 1e-6));
SYMPA_UNROLL
        for (int r = 0; r < N; ++r)
SYMPA_UNROLL
            for (int c = 0; c < N; ++c) { g.re[r][c] *= coef; g.im[r][c] *= coef; }
    }
    bool moved;
    double pow1 = 0.0, pow2 = 0.0;
    if constexpr (ADAM) {
        // this step's powers b^t: every block reads the old words here, the last block to finish stores the new ones
        pow1 = a.apows[0] * a.b1;
        pow2 = a.apows[1] * a.b2;
        double vn;
        moved = radam_row_update<N, MODEL>(z, g, a.am + ii * ROW, a.av[ii], vn, a.lr, a.b1, a.b2, a.aeps, a.wd, pow1, pow2, a.eps, st);
        if (live) {
            sympa::store_full<N>(a.am + i * ROW, g);
            a.av[i] = vn;
        }
    } else {
        moved = sympa::rsgd_row<N, MODEL>(z, g, a.lr, a.wd, a.eps, st);
    }
    if (live) {
        sympa::store_full<N>(a.table + i * ROW, z);
        if (a.zero_grads) {
SYMPA_UNROLL
            for (int e = 0; e < ROW; e += 2) *reinterpret_cast<v2d*>(a.grad + i * ROW + e) = v2d{0.0, 0.0};
        }
    }
    if (blockIdx.x == 0 && wave == 0) {
SYMPA_UNROLL
        for (int k = 0; k < FUSED_MAX_EXTRA; ++k)
            if (a.xp[k] != nullptr && lane < a.xn[k]) {
                const double p = a.xp[k][lane];
                const double gk = fma(a.xwd[k], p, coef * a.xg[k][lane]);
                if constexpr (ADAM) {       // the ordinary Adam update of a parameter without a manifold
                    const double q1 = a.xapows[k][0] * a.b1, q2 = a.xapows[k][1] * a.b2;
                    const double mk = fma(a.b1, a.xam[k][lane], (1.0 - a.b1) * gk);
                    const double vk = fma(a.b2, a.xav[k][lane], (1.0 - a.b2) * gk * gk);
                    a.xam[k][lane] = mk;
                    a.xav[k][lane] = vk;
                    a.xp[k][lane] = p - a.xlr[k] * (mk / (1.0 - q1)) / (sqrt(vk / (1.0 - q2)) + a.aeps);
                } else {
                    a.xp[k][lane] = fma(-a.xlr[k], gk, p);
                }
                if (a.zero_grads) a.xg[k][lane] = 0.0;
            }
    }
    const unsigned long long m = __ballot(live && moved);
    if (a.projected != nullptr && m != 0ull && lane == 0) atomicAdd(a.projected, (int)__popcll(m));
    if (a.status != nullptr) {
        const unsigned long long f = __ballot(st != 0 && (live || threadIdx.x == 0));
        if (f != 0ull) {
            if (st != 0) atomicOr(&a.status[0], st);
            if (lane == 0) atomicAdd(&a.status[1], (int)__popcll(f));
        }
    }
    if (ADAM || (a.max_norm > 0.0 && a.sq_in == nullptr) || a.counter != nullptr) {
        __syncthreads();       // every wave of the block has read total_s / the partials / the powers
        if (threadIdx.x == 0) {
            const unsigned done = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.x - 1) {
                __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.counter != nullptr) a.counter[0] += 1;
                if constexpr (ADAM) {       // everybody has read the old powers (a block reads them before it counts itself)
                    a.apows[0] = pow1;
                    a.apows[1] = pow2;
SYMPA_UNROLL
                    for (int k = 0; k < FUSED_MAX_EXTRA; ++k)
                        if (a.xp[k] != nullptr) { a.xapows[k][0] *= a.b1; a.xapows[k][1] *= a.b2; }
                }
            }
        }
    }
}

template <int N>
int launch_fused_step(const FusedStepArgs& a, int model, int block, hipStream_t s) {
    const unsigned grid = (unsigned)((a.rows + block - 1) / block);
    const bool up = model == SYMPA_MODEL_UPPER;
    if (a.am != nullptr) {
        if (block == 64) {
            if (up) hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_UPPER, 64, true>), dim3(grid), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_BOUNDED, 64, true>), dim3(grid), dim3(64), 0, s, a);
        } else {
            if (up) hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_UPPER, BLOCK, true>), dim3(grid), dim3(BLOCK), 0, s, a);
            else hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_BOUNDED, BLOCK, true>), dim3(grid), dim3(BLOCK), 0, s, a);
        }
    } else if (block == 64) {
        if (up) hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_UPPER, 64>), dim3(grid), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_BOUNDED, 64>), dim3(grid), dim3(64), 0, s, a);
    } else {
        if (up) hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_UPPER, BLOCK>), dim3(grid), dim3(BLOCK), 0, s, a);
        else hipLaunchKernelGGL((fused_step_kernel<N, sympa::MODEL_BOUNDED, BLOCK>), dim3(grid), dim3(BLOCK), 0, s, a);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

template <int N>
int launch_table(int op, int model, double* z, const double* g, double* out, int64_t b, double lr, double wd,
                 double eps, int32_t* projected, int32_t* status, hipStream_t s, const double* clip, double max_norm,
                 const int* gate = nullptr) {
    const unsigned grid = (unsigned)((b + BLOCK - 1) / BLOCK);
    const bool up = model == SYMPA_MODEL_UPPER;
    if (op == 3) {
        if (up) hipLaunchKernelGGL((tangent_sqnorm_kernel<N, sympa::MODEL_UPPER>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b, status);
        else hipLaunchKernelGGL((tangent_sqnorm_kernel<N, sympa::MODEL_BOUNDED>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b, status);
    } else if (op == 2) {
        if (up) hipLaunchKernelGGL((egrad2rgrad_kernel<N, sympa::MODEL_UPPER>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b);
        else hipLaunchKernelGGL((egrad2rgrad_kernel<N, sympa::MODEL_BOUNDED>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b);
    } else if (op == 0) {
        if (up) hipLaunchKernelGGL((table_update_kernel<N, sympa::MODEL_UPPER, 0>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b, lr, wd, eps, projected, status, clip, max_norm, gate);
        else hipLaunchKernelGGL((table_update_kernel<N, sympa::MODEL_BOUNDED, 0>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b, lr, wd, eps, projected, status, clip, max_norm, gate);
    } else {
        if (up) hipLaunchKernelGGL((table_update_kernel<N, sympa::MODEL_UPPER, 1>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b, lr, wd, eps, projected, status, clip, max_norm, gate);
        else hipLaunchKernelGGL((table_update_kernel<N, sympa::MODEL_BOUNDED, 1>), dim3(grid), dim3(BLOCK), 0, s, z, g, out, b, lr, wd, eps, projected, status, clip, max_norm, gate);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

// n = 9..16 (siegel_table_rolled.hip)
int launch_table_rolled(int op, int n, int model, double* z, const double* g, double* out, int64_t b, double lr, double wd,
                        double eps, int32_t* projected, int32_t* status, hipStream_t s, const double* clip, double max_norm,
                        const int* gate = nullptr);
// n = 9..16, op 1 (RSGD step) and 2 (egrad2rgrad) with sixteen lanes per row (siegel_table_coop_{upper,bounded}.hip);
// `outside`: device word that receives the number of rows that left the eps-interior (op 1)
int launch_table_coop_upper(int op, int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps,
                            const double* clip, double max_norm, int* outside, hipStream_t s);
int launch_table_coop_bounded(int op, int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps,
                              const double* clip, double max_norm, int* outside, hipStream_t s);
// n = 7, 8 with eight lanes per row (siegel_table_half_{upper,bounded}.hip)
int launch_table_half_upper(int op, int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps,
                            const double* clip, double max_norm, int* outside, hipStream_t s);
int launch_table_half_bounded(int op, int n, double* z, const double* g, double* out, int64_t b, double lr, double wd, double eps,
                              const double* clip, double max_norm, int* outside, hipStream_t s);

}  // namespace sympa_hip
