// gfx950 kernel + C-ABI of the SPD model (spd_math.hpp): runtime-n, per-lane scratch, 64-thread blocks.
#include "siegel_common.hpp"
#include "spd_math.hpp"

namespace {
using namespace sympa_hip;

__global__ __launch_bounds__(64) void spd_dist_kernel(const DistArgs a, const int n) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
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
    const int64_t row = (int64_t)n * n;
    sympa::SpdWork w;
    double d = sympa::spd_pair_distance(w, a.base1 + r1 * row, a.base2 + r2 * row, n, st);
    if (st & sympa::ST_BAD_INDEX) d = __builtin_nan("");
    if (a.scale != nullptr) d *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);
    if (live) a.out[i] = d;
    if (a.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {
            if (flagged) atomicOr(&a.status[0], st);
            if ((threadIdx.x & 63) == 0) atomicAdd(&a.status[1], (int)__popcll(m));
        }
    }
}

int launch_spd(const DistArgs& a, int n, void* stream) {
    if (a.b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
    if (a.b == 0) return 0;
    if (a.base1 == nullptr || a.base2 == nullptr || a.out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (n < 1 || n > sympa::SPD_MAX_N) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd: dims outside [1, 16]");
    if (a.num_rows > (int64_t)0x7fffffff) return fail(SYMPA_ERR_BAD_ARG, "more than 2^31-1 table rows");
    hipLaunchKernelGGL(spd_dist_kernel, dim3((unsigned)((a.b + 63) / 64)), dim3(64), 0,
                       reinterpret_cast<hipStream_t>(stream), a, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

int sympa_spd_dist_fwd(const double* x, const double* y, int64_t b, int n, double* out, int32_t* status, void* stream) {
    DistArgs a;
    std::memset(&a, 0, sizeof(a));
    a.base1 = x;
    a.base2 = y;
    a.b = b;
    a.num_rows = b;
    a.inv_scale_coef = 1.0;
    a.out = out;
    a.status = status;
    return launch_spd(a, n, stream);
}

int sympa_spd_model_forward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                            const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale, double scale_coef,
                            double* out, int32_t* status, void* stream) {
    if (b > 0 && (src == nullptr || dst == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null index buffer");
    if (num_rows <= 0 && b > 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    DistArgs a;
    std::memset(&a, 0, sizeof(a));
    a.base1 = table;
    a.base2 = table;
    a.idx1 = src;
    a.idx2 = dst;
    a.idx1_stride = src_stride;
    a.idx2_stride = dst_stride;
    a.num_rows = num_rows;
    a.b = b;
    a.scale = scale;
    a.inv_scale_coef = 1.0 / scale_coef;
    a.out = out;
    a.status = status;
    return launch_spd(a, n, stream);
}

}  // extern "C"
