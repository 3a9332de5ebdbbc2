// gfx950 kernels + C-ABI for the Siegel-distance hot path (include/sympa_hip.h).
//
// Mapping: ONE PAIR PER LANE.  A wavefront handles 64 independent pairs; the
// complex n x n matrices of a pair live entirely in that lane's VGPRs (siegel_math.hpp unrolls to
// static register indices), so the factorisations, solves and the Jacobi iteration need no
// cross-lane traffic at all and no lane idles in an elementwise stage.  The only wave-level
// operation is the ballot that ends the Jacobi loop when all 64 pairs have converged.
// The workload is VALU-fp64 bound: the table rows come out of L2 / Infinity Cache.
#include <atomic>
#include "siegel_dist_kernel.hpp"
#include "siegel_math_generic.hpp"

namespace sympa_hip {

thread_local char g_err[256] = "";
char* last_error_buffer() { return g_err; }
int fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int validate(const DistArgs& a, int model, int n) {
    if (a.b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
    if (a.base1 == nullptr || a.base2 == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (a.metric < SYMPA_METRIC_RIEM || a.metric > SYMPA_METRIC_WSUM) return fail(SYMPA_ERR_BAD_ARG, "unknown metric");
    if (a.metric == SYMPA_METRIC_WSUM && a.metric_w == nullptr)
        return fail(SYMPA_ERR_BAD_ARG, "metric wsum needs metric_w");
    if (!(a.inv_eps > 0.0) || !(a.inv_eps < 1e300)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    if (a.b > (int64_t)0x7fffffff * 64) return fail(SYMPA_ERR_BAD_ARG, "batch too large for one launch");
    if (a.num_rows > (int64_t)0x7fffffff) return fail(SYMPA_ERR_BAD_ARG, "more than 2^31-1 table rows");
    // only the LDS-DMA / staged gathers of dims <= 4 address rows with 32-bit byte offsets
    if (n <= 4 && a.num_rows * 16 * n * n >= ((int64_t)1 << 32))
        return fail(SYMPA_ERR_BAD_ARG, "dims <= 4: the table (or the batch of pre-gathered points) is limited to 4 GiB "
                                       "per launch (32-bit row offsets in the gather); split the call");
    return 0;
}

}  // namespace sympa_hip

namespace {
using namespace sympa_hip;

// ---------------------------------------------------------------------------------------------
// Backward.  Same lane-per-pair mapping; the forward quantities are recomputed.
// SCATTER: the two gradient rows of a pair are added into grad_table[src], grad_table[dst] (the dense
// [N,2,n,n] tensor autograd's embedding backward would build).  The per-lane rows are first transposed
// through the wave's LDS tile so that one atomic wave-instruction covers whole contiguous rows
// (64 / 2n^2 rows x 16n^2 B), the access shape the fp64 atomics need to run at rate.
// ---------------------------------------------------------------------------------------------
// Runtime-n fallback (9 <= n <= 16): per-lane matrices in scratch, direct loads, 64-thread blocks.
__global__ __launch_bounds__(64) void siegel_dist_generic_kernel(const DistArgs a, const int n, const int model) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
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
    const int64_t row = 2 * (int64_t)n * n;
    sympa::GenericWork w;
    double* vv = (a.vvd != nullptr && live) ? a.vvd + i * n : nullptr;
    double d = sympa::pair_distance_generic(w, a.base1 + r1 * row, a.base2 + r2 * row, n, model, a.metric, a.metric_w,
                                            a.inv_eps, vv, st);
    if (st & sympa::ST_BAD_INDEX) d = __builtin_nan("");
    if (a.scale != nullptr) d *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);
    if (live) {
        if (a.ap_cols > 0) ap_store(a, i, r1, r2, d);
        else a.out[i] = d;
    }
    if (a.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {
            if (flagged) atomicOr(&a.status[0], st);
            if ((threadIdx.x & 63) == 0) atomicAdd(&a.status[1], (int)__popcll(m));
        }
    }
}

int launch(const DistArgs& a, int n, int model, void* stream) {
    if (a.b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
    if (a.b == 0) return 0;
    if (a.base1 == nullptr || a.base2 == nullptr || a.out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (a.metric < SYMPA_METRIC_RIEM || a.metric > SYMPA_METRIC_WSUM) return fail(SYMPA_ERR_BAD_ARG, "unknown metric");
    if (a.metric == SYMPA_METRIC_WSUM && a.metric_w == nullptr)
        return fail(SYMPA_ERR_BAD_ARG, "metric wsum needs metric_w");
    if (!(a.inv_eps > 0.0) || !(a.inv_eps < 1e300)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    if (a.b > (int64_t)0x7fffffff * 64) return fail(SYMPA_ERR_BAD_ARG, "batch too large for one launch");
    if (a.num_rows > (int64_t)0x7fffffff) return fail(SYMPA_ERR_BAD_ARG, "more than 2^31-1 table rows");
    if (n <= 4 && a.num_rows * 16 * n * n >= ((int64_t)1 << 32))
        return fail(SYMPA_ERR_BAD_ARG, "tables of dims <= 4 are limited to 4 GiB (32-bit row offsets in the gather)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((a.flags & SYMPA_FLAG_COOP) && (n == 7 || n == 8)) return launch_siegel_coop_half(a, n, model, s);   // A/B only: eight lanes per pair
    if ((a.flags & SYMPA_FLAG_COOP) && n == 6) return launch_siegel_coop(a, n, model, s);                 // A/B only: sixteen
    if (n >= 7 && n <= 8 && model == SYMPA_MODEL_UPPER && a.ap_cols == 0 && a.vvd == nullptr && a.batch_counter == nullptr &&
        !(a.flags & (SYMPA_FLAG_ANY_ORDER | SYMPA_FLAG_GENERIC))) {
        // round 5: persistent waves, prefetched tile head, in-place difference (siegel_packed_kernel.hpp dense_forward_kernel);
        // the one-launch kernel of siegel_dist_kernel.hpp remains for the vector-valued distance, the all-pairs mode and the
        // bounded model (whose large calls go through the packed table)
        if (a.idx1 == nullptr && a.base1 == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
        PackedArgs p;
        std::memset(&p, 0, sizeof(p));
        p.pack = a.base1;
        p.base2 = a.base2;
        p.identity = a.idx1 == nullptr ? 1 : 0;
        p.num_rows = a.num_rows;
        p.idx1[0] = a.idx1;
        p.idx2[0] = a.idx2;
        p.out[0] = a.out;
        p.b[0] = a.b;
        p.stride1 = a.idx1_stride;
        p.stride2 = a.idx2_stride;
        p.metric_w = a.metric_w;
        p.scale = a.scale;
        p.inv_scale_coef = a.inv_scale_coef;
        p.inv_eps = a.inv_eps;
        p.status = a.status;
        p.metric = a.metric;
        p.num_batches = 1;
        return launch_dense_persistent(p, n, s);
    }
    switch (n) {
        case 1: return launch_n<1>(a, model, s);
        case 2: return launch_n<2>(a, model, s);
        case 3: return launch_n<3>(a, model, s);
        case 4: return launch_n<4>(a, model, s);
        case 5: case 6: case 7: case 8: return launch_dist_big(a, n, model, s);
        default: break;
    }
    if (n > SYMPA_MAX_DIMS && n <= sympa::GENERIC_MAX_N) {
        if (!(a.flags & SYMPA_FLAG_GENERIC) && !instance_fallback(SYMPA_FAMILY_SIEGEL_FWD, model, n))
            return launch_siegel_coop(a, n, model, s);
        hipLaunchKernelGGL(siegel_dist_generic_kernel, dim3((unsigned)((a.b + 63) / 64)), dim3(64), 0, s, a, n, model);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
        return 0;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "dims outside [1, SYMPA_MAX_DIMS_GENERIC]");
}

int launch_multi(const double* table, int64_t num_rows, int n, const int64_t* const* triplets, int64_t stride,
                 const int64_t* b, int cnt, int model, int metric, const double* metric_w, double eps,
                 const double* scale, double scale_coef, double* const* out, int32_t* status, int flags, void* stream) {
    if (table == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (num_rows <= 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (metric < SYMPA_METRIC_RIEM || metric > SYMPA_METRIC_WSUM) return fail(SYMPA_ERR_BAD_ARG, "unknown metric");
    if (metric == SYMPA_METRIC_WSUM && metric_w == nullptr) return fail(SYMPA_ERR_BAD_ARG, "metric wsum needs metric_w");
    if (!(eps > 0.0) || !(1.0 / eps < 1e300)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    if (num_rows > (int64_t)0x7fffffff) return fail(SYMPA_ERR_BAD_ARG, "more than 2^31-1 table rows");
    if (n <= 4 && num_rows * 16 * n * n >= ((int64_t)1 << 32))
        return fail(SYMPA_ERR_BAD_ARG, "tables of dims <= 4 are limited to 4 GiB (32-bit row offsets in the gather)");
    MultiArgs m;
    std::memset(&m, 0, sizeof(m));
    m.c.base1 = table;
    m.c.base2 = table;
    m.c.idx1_stride = stride;
    m.c.idx2_stride = stride;
    m.c.num_rows = num_rows;
    m.c.metric_w = metric_w;
    m.c.scale = scale;
    m.c.inv_scale_coef = 1.0 / scale_coef;
    m.c.inv_eps = 1.0 / eps;
    m.c.status = status;
    m.c.metric = metric;
    m.c.flags = flags;
    uint64_t blocks = 0;
    int k = 0;
    const int fb = fwd_block(n);
    for (int i = 0; i < cnt; ++i) {
        if (b[i] < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
        if (b[i] == 0) continue;
        if (triplets[i] == nullptr || out[i] == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
        blocks += (uint64_t)((b[i] + fb - 1) / fb);
        if (blocks > 0x7fffffffull) return fail(SYMPA_ERR_BAD_ARG, "batches too large for one launch");
        m.trip[k] = triplets[i];
        m.out[k] = out[i];
        m.b[k] = b[i];
        m.blk_end[k] = (unsigned)blocks;
        ++k;
    }
    if (k == 0) return 0;
    for (int i = k; i < SYMPA_MAX_FUSED_BATCHES; ++i) m.blk_end[i] = (unsigned)blocks;   // the search never lands there
    m.num_batches = k;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n >= 7 && n <= 8 && model == SYMPA_MODEL_UPPER && !(flags & (SYMPA_FLAG_ANY_ORDER | SYMPA_FLAG_GENERIC))) {
        PackedArgs p;                       // the list form of the persistent dense forward (see launch())
        std::memset(&p, 0, sizeof(p));
        p.pack = table;
        p.base2 = table;
        p.num_rows = num_rows;
        for (int i = 0; i < k; ++i) {
            p.idx1[i] = m.trip[i];
            p.idx2[i] = m.trip[i] + 1;
            p.out[i] = m.out[i];
            p.b[i] = m.b[i];
        }
        p.stride1 = stride;
        p.stride2 = stride;
        p.metric_w = metric_w;
        p.scale = scale;
        p.inv_scale_coef = 1.0 / scale_coef;
        p.inv_eps = 1.0 / eps;
        p.status = status;
        p.metric = metric;
        p.num_batches = k;
        return launch_dense_persistent(p, n, s);
    }
    switch (n) {
        case 1: return launch_multi_n<1>(m, (unsigned)blocks, model, s);
        case 2: return launch_multi_n<2>(m, (unsigned)blocks, model, s);
        case 3: return launch_multi_n<3>(m, (unsigned)blocks, model, s);
        case 4: return launch_multi_n<4>(m, (unsigned)blocks, model, s);
        case 5: case 6: case 7: case 8: return launch_multi_big(m, (unsigned)blocks, n, model, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "fused batches need dims <= SYMPA_MAX_DIMS");
}

}  // namespace

namespace sympa_hip {
namespace {
std::atomic<unsigned char> g_instance_fallback[SYMPA_NUM_FAMILIES][2][SYMPA_MAX_DIMS_GENERIC + 1];
}
bool instance_fallback(int family, int model, int n) {
    if (family < 0 || family >= SYMPA_NUM_FAMILIES || model < 0 || model > 1 || n < 0 || n > SYMPA_MAX_DIMS_GENERIC) return false;
    return g_instance_fallback[family][model][n].load(std::memory_order_relaxed) != 0;
}
}  // namespace sympa_hip

extern "C" {

int sympa_set_instance_fallback(int family, int model, int n, int on) {
    if (family < 0 || family >= SYMPA_NUM_FAMILIES || model < 0 || model > 1 || n < 1 || n > SYMPA_MAX_DIMS_GENERIC)
        return sympa_hip::fail(SYMPA_ERR_BAD_ARG, "no such kernel instantiation");
    sympa_hip::g_instance_fallback[family][model][n].store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

int sympa_get_instance_fallback(int family, int model, int n) { return sympa_hip::instance_fallback(family, model, n) ? 1 : 0; }

const char* sympa_version(void) { return "sympa_hip 0.1.0 (gfx950)"; }
const char* sympa_last_error(void) { return sympa_hip::last_error_buffer(); }
int sympa_max_dims(void) { return SYMPA_MAX_DIMS; }

int sympa_siegel_dist_fwd(const double* z1, const double* z2, int64_t b, int n, int model, int metric,
                          const double* metric_w, double eps, double* out, double* vvd_out, int32_t* status,
                          int flags, void* stream) {
    DistArgs a;
    std::memset(&a, 0, sizeof(a));
    a.base1 = z1;
    a.base2 = z2;
    a.b = b;
    a.num_rows = b;
    a.metric_w = metric_w;
    a.inv_scale_coef = 1.0;
    a.inv_eps = 1.0 / eps;
    a.out = out;
    a.vvd = vvd_out;
    a.status = status;
    a.metric = metric;
    a.flags = flags;
    return launch(a, n, model, stream);
}

int sympa_model_forward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                        const int64_t* dst, int64_t dst_stride, int64_t b, int model, int metric,
                        const double* metric_w, double eps, const double* scale, double scale_coef, double* out,
                        int32_t* status, int flags, void* stream) {
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
    a.metric_w = metric_w;
    a.scale = scale;
    a.inv_scale_coef = 1.0 / scale_coef;
    a.inv_eps = 1.0 / eps;
    a.out = out;
    a.vvd = nullptr;
    a.status = status;
    a.metric = metric;
    a.flags = flags;
    return launch(a, n, model, stream);
}

int sympa_model_forward_batches(const double* table, int64_t num_rows, int n, const int64_t* const* triplets,
                                int64_t stride, const int64_t* b, int num_batches, int model, int metric,
                                const double* metric_w, double eps, const double* scale, double scale_coef,
                                double* const* out, int32_t* status, int flags, void* const* streams, int num_streams) {
    if (num_batches < 0 || num_streams < 1 || streams == nullptr) return fail(SYMPA_ERR_BAD_ARG, "bad batch / stream list");
    if (num_batches > 0 && (triplets == nullptr || b == nullptr || out == nullptr))
        return fail(SYMPA_ERR_BAD_ARG, "null batch list");
    if (stride < 2) return fail(SYMPA_ERR_BAD_ARG, "triplet stride must be >= 2");
    if ((flags & SYMPA_FLAG_FUSE) && n >= 1 && n <= SYMPA_MAX_DIMS) {
        // groups of up to SYMPA_MAX_FUSED_BATCHES batches per launch; group g goes to streams[g % num_streams]
        int group = 0;
        for (int i0 = 0; i0 < num_batches; i0 += SYMPA_MAX_FUSED_BATCHES, ++group) {
            const int cnt = num_batches - i0 < SYMPA_MAX_FUSED_BATCHES ? num_batches - i0 : SYMPA_MAX_FUSED_BATCHES;
            const int rc = launch_multi(table, num_rows, n, triplets + i0, stride, b + i0, cnt, model, metric, metric_w,
                                        eps, scale, scale_coef, out + i0, status, flags, streams[group % num_streams]);
            if (rc != 0) return rc;
        }
        return 0;
    }
    for (int i = 0; i < num_batches; ++i) {
        const int rc = sympa_model_forward(table, num_rows, n, triplets[i], stride, triplets[i] + 1, stride, b[i], model,
                                           metric, metric_w, eps, scale, scale_coef, out[i], status, flags,
                                           streams[i % num_streams]);
        if (rc != 0) return rc;
    }
    return 0;
}

int sympa_all_pairs_dist(const double* table, int64_t num_rows, int n, int64_t row_begin, int64_t row_count, int model,
                         int metric, const double* metric_w, double eps, const double* scale, double scale_coef,
                         double* out, int32_t* status, int flags, void* stream) {
    if (num_rows <= 0 || row_begin < 0 || row_count < 0 || row_begin + row_count > num_rows)
        return fail(SYMPA_ERR_BAD_ARG, "row block outside the table");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    DistArgs a;
    std::memset(&a, 0, sizeof(a));
    a.base1 = table;
    a.base2 = table;
    a.num_rows = num_rows;
    a.b = row_count * num_rows;
    a.metric_w = metric_w;
    a.scale = scale;
    a.inv_scale_coef = 1.0 / scale_coef;
    a.inv_eps = 1.0 / eps;
    a.out = out;
    a.status = status;
    a.metric = metric;
    a.flags = flags;
    a.ap_cols = num_rows;
    a.ap_row0 = row_begin;
    // the full matrix: d(i, j) = d(j, i), evaluate the pairs i <= j only and store each value twice (dims >= 3: below that
    // a pair costs less than the scattered mirror store)
    if (row_begin == 0 && row_count == num_rows && n >= 3 && !(flags & SYMPA_FLAG_NO_SYMMETRY)) {
        a.ap_sym = 1;
        a.b = num_rows * (num_rows + 1) / 2;
    }
    return launch(a, n, model, stream);
}

}  // extern "C"
