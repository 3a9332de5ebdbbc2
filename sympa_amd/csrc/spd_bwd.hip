// gfx950 kernels + C-ABI of the SPD model's training path (spd_math_bwd.hpp): backward of the affine-invariant
// distance with the fused AverageDistortionLoss, and the optimiser-side row operations (egrad2rgrad, projx, RSGD step).
// Sixteen lanes per pair / per row for n >= 3 (spd_coop_bwd_kernel.hpp, spd_coop_table.hpp); the one-pair-per-lane kernels
// below (runtime n <= 16, per-lane scratch) serve n <= 2, projx, and SYMPA_FLAG_GENERIC / SYMPA_SPD_TABLE_GENERIC=1 (A/B).
// PARITY UNPINNED with respect to geoopt (absent); pinned by mpmath finite differences and autograd through the oracle.
#include <cstdlib>

#include "siegel_common.hpp"
#include "spd_math_bwd.hpp"
#include "spd_coop_bwd3_kernel.hpp"

namespace {
using namespace sympa_hip;


__global__ __launch_bounds__(64) void spd_bwd_kernel(const SpdBwdArgs a, const int n) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = i < a.b;
    const int64_t ii = live ? i : a.b - 1;
    int st = 0;
    int64_t r1 = ii, r2 = ii;
    if (a.src != nullptr) {
        r1 = a.src[ii * a.src_stride];
        r2 = a.dst[ii * a.dst_stride];
        if (r1 < 0 || r1 >= a.num_rows || r2 < 0 || r2 >= a.num_rows) { st |= sympa::ST_BAD_INDEX; r1 = 0; r2 = 0; }
    }
    const int nn = n * n;
    sympa::SpdBwdWork w;
    double* gx = a.gx + ii * nn;      // tail lanes recompute the last pair and do not store
    double* gy = a.gy + ii * nn;
    double lgx[sympa::SPD_MAX_N * sympa::SPD_MAX_N], lgy[sympa::SPD_MAX_N * sympa::SPD_MAX_N];
    const double dist = sympa::spd_pair_backward(w, a.x + r1 * nn, a.y + r2 * nn, n, lgx, lgy, st);
    double sc = 1.0;
    bool sc_active = false;
    if (a.scale != nullptr) {
        const double raw = a.scale[0] * a.inv_scale_coef;
        sc_active = raw > 0.1;
        sc = sc_active ? raw : 0.1;
    }
    const bool bad = (st & sympa::ST_BAD_INDEX) != 0;
    double go = 0.0, loss_i = 0.0;
    if (a.graph_dist != nullptr) {            // AverageDistortionLoss (losses.py:10-19): sum |(d/g)^2 - 1|
        const double gd = live ? a.graph_dist[i] : 1.0;
        const double ratio = dist * sc / gd;
        const double e = ratio * ratio - 1.0;
        loss_i = (live && !bad) ? fabs(e) * a.loss_scale : 0.0;
        go = (e > 0.0 ? 1.0 : (e < 0.0 ? -1.0 : 0.0)) * 2.0 * ratio / gd * a.loss_scale;
    } else if (a.go != nullptr) {
        go = live ? a.go[i] : 0.0;
    }
    if (!live || bad) go = 0.0;
    if (live) {
        const double f = go * sc;
        for (int k = 0; k < nn; ++k) { gx[k] = f * lgx[k]; gy[k] = f * lgy[k]; }
        if (a.out != nullptr) a.out[i] = bad ? __builtin_nan("") : dist * sc;
    }
    if (a.gscale != nullptr && a.scale != nullptr) {
        double v = (live && sc_active) ? go * dist * a.inv_scale_coef : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (threadIdx.x == 0 && v != 0.0) atomicAdd(a.gscale, v);
    }
    if (a.loss != nullptr && a.graph_dist != nullptr) {
        double v = loss_i;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (threadIdx.x == 0 && v != 0.0) atomicAdd(a.loss, v);
    }
    if (a.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {
            if (flagged) atomicOr(&a.status[0], st);
            if (threadIdx.x == 0) atomicAdd(&a.status[1], (int)__popcll(m));
        }
    }
}

// OP 0: out = projx(x)   1: table <- RSGD step in place   2: out = egrad2rgrad(x, g)
__global__ __launch_bounds__(64) void spd_table_kernel(const int op, double* x, const double* g, double* out, const int64_t b,
                                                       const int n, const double lr, const double wd, const double* clip,
                                                       const double max_norm, int32_t* projected, int32_t* status) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = i < b;
    const int64_t ii = live ? i : b - 1;
    const int nn = n * n;
    sympa::SpdRowWork w;
    int st = 0;
    bool moved = false;
    double row[sympa::SPD_MAX_N * sympa::SPD_MAX_N];
    if (op == 0) {
        moved = sympa::spd_row_projx(w, x + ii * nn, n, row, st);
        if (live) for (int k = 0; k < nn; ++k) out[i * nn + k] = row[k];
    } else if (op == 2) {
        sympa::spd_row_egrad2rgrad(w, x + ii * nn, g + ii * nn, n, row);
        if (live) for (int k = 0; k < nn; ++k) out[i * nn + k] = row[k];
    } else {
        const double coef = (clip != nullptr) ? fmin(1.0, max_norm / (sqrt(clip[0]) + 1e-6)) : 1.0;
        for (int k = 0; k < nn; ++k) row[k] = x[ii * nn + k];
        sympa::spd_row_rsgd(w, row, g + ii * nn, n, lr, wd, coef, st);
        if (live) for (int k = 0; k < nn; ++k) x[i * nn + k] = row[k];
    }
    const unsigned long long m = __ballot(live && moved);
    if (projected != nullptr && m != 0ull && threadIdx.x == 0) atomicAdd(projected, (int)__popcll(m));
    if (status != nullptr) {
        const unsigned long long f = __ballot(live && st != 0);
        if (f != 0ull) {
            if (live && st != 0) atomicOr(&status[0], st);
            if (threadIdx.x == 0) atomicAdd(&status[1], (int)__popcll(f));
        }
    }
}

int launch_spd_table(int op, double* x, const double* g, double* out, int64_t b, int n, double lr, double wd,
                     const double* clip, double max_norm, int32_t* projected, int32_t* status, void* stream) {
    if (b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative row count");
    if (b == 0) return 0;
    if (x == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (n < 1 || n > sympa::SPD_MAX_N) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd: dims outside [1, 16]");
    // SYMPA_SPD_TABLE_GENERIC=1 keeps the one-row-per-lane kernel for A/B measurements (tools/spd_time.py)
    static const bool generic = std::getenv("SYMPA_SPD_TABLE_GENERIC") != nullptr;
    if (op != 0 && n >= SPD_COOP_BWD_MIN_N && !generic && !instance_fallback(SYMPA_FAMILY_SPD_TABLE, 0, n))
        launch_spd_coop_table(op, n, x, g, out, b, lr, wd, clip, max_norm, status, reinterpret_cast<hipStream_t>(stream));
    else
        hipLaunchKernelGGL(spd_table_kernel, dim3((unsigned)((b + 63) / 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                           op, x, g, out, b, n, lr, wd, clip, max_norm, projected, status);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

static int spd_backward_impl(const double* x, const double* y, int64_t num_rows, int n, const int64_t* src,
                            int64_t src_stride, const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale,
                            double scale_coef, const double* grad_out, const double* graph_dist, double loss_scale,
                            double* loss, double* grad_x_rows, double* grad_y_rows, double* grad_scale, double* out,
                            int32_t* status, int flags, void* stream, double* grad_table, void* workspace,
                            int64_t workspace_bytes) {
    if (b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
    if (b == 0) return 0;
    if (x == nullptr || y == nullptr || (grad_table == nullptr && (grad_x_rows == nullptr || grad_y_rows == nullptr)))
        return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (instance_fallback(SYMPA_FAMILY_SPD_BWD, 0, n)) flags |= SYMPA_FLAG_GENERIC;
    if (grad_table != nullptr && (n < SPD_COOP_BWD_MIN_N || src == nullptr || (flags & SYMPA_FLAG_GENERIC)))
        return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd: the in-kernel scatter needs n >= 3, index lists and the sixteen-lanes kernel "
                                                "(not SYMPA_FLAG_GENERIC / an instance fallback): use sympa_spd_backward_rows + "
                                                "sympa_scatter_add_flat_rows");
    if ((src == nullptr) != (dst == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "give both index lists or neither");
    if (src != nullptr && num_rows <= 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (grad_out == nullptr && graph_dist == nullptr) return fail(SYMPA_ERR_BAD_ARG, "need grad_out or graph_dist");
    if (n < 1 || n > sympa::SPD_MAX_N) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd: dims outside [1, 16]");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    SpdBwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.src = src; a.dst = dst; a.src_stride = src_stride; a.dst_stride = dst_stride;
    a.num_rows = num_rows; a.b = b; a.scale = scale; a.inv_scale_coef = 1.0 / scale_coef;
    a.go = grad_out; a.graph_dist = graph_dist; a.loss_scale = loss_scale; a.loss = loss;
    a.gtab = grad_table; a.gx = grad_x_rows; a.gy = grad_y_rows; a.gscale = grad_scale; a.out = out; a.status = status;
    const dim3 grid((unsigned)((b + 63) / 64));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // with a workspace: the three-phase kernel (eigenvectors one pair per lane by inverse iteration), where instantiated
    if (workspace != nullptr && workspace_bytes > 0 && !(flags & (SYMPA_FLAG_COOP | SYMPA_FLAG_GENERIC))) {
        int rc3 = 0;
        if (launch_spd_bwd3(a, n, workspace, workspace_bytes, s, &rc3)) return rc3;
    }
    // default: the QL of two rounds run together (8 pairs per wave and step) once the batch fills the chip that way;
    // SYMPA_FLAG_COOP forces the single-round kernel (A/B), SYMPA_FLAG_GENERIC the one-lane-per-pair kernel
    if (!(flags & (SYMPA_FLAG_COOP | SYMPA_FLAG_GENERIC)) && n >= 4 && b >= 8192 &&
        (n <= 10 ? launch_spd_bwd_coop2_lo(a, n, s) : launch_spd_bwd_coop2_hi(a, n, s))) {
    } else if (n >= SPD_COOP_BWD_MIN_N && !(flags & SYMPA_FLAG_GENERIC)) {
        if (n == 16) hipLaunchKernelGGL(spd_coop_bwd_kernel<16>, spd_coop_bwd_grid(a.b, 16), dim3(64), 0, s, a, spd_coop::coop_rounds(a.b, spd_coop_bwd_waves<16>()));
        else if (n >= 12) launch_spd_coop_bwd_hi(a, n, grid, s);
        else launch_spd_coop_bwd_lo(a, n, grid, s);
    } else {
        hipLaunchKernelGGL(spd_bwd_kernel, grid, dim3(64), 0, s, a, n);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}


int sympa_spd_backward_rows(const double* x, const double* y, int64_t num_rows, int n, const int64_t* src,
                            int64_t src_stride, const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale,
                            double scale_coef, const double* grad_out, const double* graph_dist, double loss_scale,
                            double* loss, double* grad_x_rows, double* grad_y_rows, double* grad_scale, double* out,
                            int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream) {
    return spd_backward_impl(x, y, num_rows, n, src, src_stride, dst, dst_stride, b, scale, scale_coef, grad_out, graph_dist,
                             loss_scale, loss, grad_x_rows, grad_y_rows, grad_scale, out, status, flags, stream, nullptr,
                             workspace, workspace_bytes);
}

int64_t sympa_spd_backward_workspace_bytes(int64_t b, int n) {
    if (b <= 0 || n < 9 || n > 16) return 0;               // the three-kernel backward is instantiated for n = 9..16
    return spd_bwd3_workspace_bytes(b, n);
}

int sympa_spd_loss_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                            const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale, double scale_coef,
                            const double* grad_out, const double* graph_dist, double loss_scale, double* loss,
                            double* grad_table, double* grad_scale, double* out, int32_t* status, void* workspace,
                            int64_t workspace_bytes, int flags, void* stream) {
    if (b > 0 && grad_table == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null gradient table");
    return spd_backward_impl(table, table, num_rows, n, src, src_stride, dst, dst_stride, b, scale, scale_coef, grad_out,
                             graph_dist, loss_scale, loss, nullptr, nullptr, grad_scale, out, status, flags, stream, grad_table,
                             workspace, workspace_bytes);
}

int sympa_spd_egrad2rgrad(const double* x, const double* u, int64_t b, int n, double* out, void* stream) {
    if (b > 0 && (u == nullptr || out == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    return launch_spd_table(2, const_cast<double*>(x), u, out, b, n, 0.0, 0.0, nullptr, 0.0, nullptr, nullptr, stream);
}

int sympa_spd_projx(const double* x, int64_t b, int n, double* out, int32_t* projected_count, int32_t* status, void* stream) {
    if (b > 0 && out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    return launch_spd_table(0, const_cast<double*>(x), nullptr, out, b, n, 0.0, 0.0, nullptr, 0.0, projected_count, status, stream);
}

int sympa_spd_rsgd_step(double* table, const double* grad, int64_t num_rows, int n, double lr, double weight_decay,
                        const double* total_sqnorm, double max_norm, int32_t* status, void* stream) {
    if (num_rows > 0 && grad == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null gradient");
    if (total_sqnorm != nullptr && !(max_norm > 0.0)) return fail(SYMPA_ERR_BAD_ARG, "max_norm must be > 0");
    return launch_spd_table(1, table, grad, nullptr, num_rows, n, lr, weight_decay, total_sqnorm, max_norm, nullptr, status, stream);
}

}  // extern "C"
