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
