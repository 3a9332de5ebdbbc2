// C-ABI of the packed indexed forward, dims 5..8 (siegel_packed_kernel.hpp): sympa_table_pack_bytes / sympa_table_pack /
// sympa_model_forward_packed / sympa_model_forward_batches_packed.
// Compiled like siegel_dist_big.hip without the pre-RA machine scheduler (__graft_entry__.py: the fully unrolled dims-8 bodies keep
// the source order).
#include "siegel_packed_kernel.hpp"
#include "table_digest.hpp"

namespace {
using namespace sympa_hip;

int pack_row_doubles(int n, int model) {
    const int tri = n * (n + 1) / 2, low = n * (n - 1) / 2;
    const int len = 2 * tri + n + (model == SYMPA_MODEL_UPPER ? low : 2 * low);
    return 2 * ((len + 1) / 2);
}

int cu_count() {
    static int cached[16] = {0};          // by device ordinal: the attribute query is not free in front of a ~100 us kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int cus = cached[dev];
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev] = cus;
    }
    return cus;
}

template <int N>
int pack_n(const double* table, int64_t num_rows, int model, double* pack, int32_t* status, const unsigned* guard, hipStream_t s) {
    return model == SYMPA_MODEL_UPPER ? launch_table_pack<N, sympa::MODEL_UPPER>(table, num_rows, pack, status, guard, s)
                                      : launch_table_pack<N, sympa::MODEL_BOUNDED>(table, num_rows, pack, status, guard, s);
}

int pack_any(const double* table, int64_t num_rows, int n, int model, void* pack, int64_t pack_bytes, int32_t* status,
             const unsigned* guard, hipStream_t s) {
    if (!packed_dims_ok(n)) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "packed table: dims 5..8");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (table == nullptr || num_rows <= 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (pack == nullptr || pack_bytes < sympa_table_pack_bytes(num_rows, n, model) || (reinterpret_cast<uintptr_t>(pack) & 15))
        return fail(SYMPA_ERR_BAD_ARG, "packed table: a 16-byte aligned buffer of sympa_table_pack_bytes(num_rows, n, model) bytes");
    double* p = reinterpret_cast<double*>(pack);
    switch (n) {
        case 5: return pack_n<5>(table, num_rows, model, p, status, guard, s);
        case 6: return pack_n<6>(table, num_rows, model, p, status, guard, s);
        case 7: return pack_n<7>(table, num_rows, model, p, status, guard, s);
        case 8: return pack_n<8>(table, num_rows, model, p, status, guard, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "packed table: dims 5..8");
}

template <int N>
int forward_n(const PackedArgs& a, unsigned grid, int model, hipStream_t s) {
    return model == SYMPA_MODEL_UPPER ? launch_packed_forward<N, sympa::MODEL_UPPER>(a, grid, s)
                                      : launch_packed_forward<N, sympa::MODEL_BOUNDED>(a, grid, s);
}

int check_common(const void* pack, int64_t pack_bytes, int64_t num_rows, int n, int model, int metric, const double* metric_w,
                 double eps, const double* scale, double scale_coef) {
    if (!packed_dims_ok(n)) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "packed forward: dims 5..8");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (num_rows <= 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (num_rows > (int64_t)0x7fffffff) return fail(SYMPA_ERR_BAD_ARG, "more than 2^31-1 table rows");
    if (pack == nullptr || pack_bytes < sympa_table_pack_bytes(num_rows, n, model) || (reinterpret_cast<uintptr_t>(pack) & 15))
        return fail(SYMPA_ERR_BAD_ARG, "packed table: a 16-byte aligned buffer of sympa_table_pack_bytes(num_rows, n, model) bytes");
    if (metric < SYMPA_METRIC_RIEM || metric > SYMPA_METRIC_WSUM) return fail(SYMPA_ERR_BAD_ARG, "unknown metric");
    if (metric == SYMPA_METRIC_WSUM && metric_w == nullptr) return fail(SYMPA_ERR_BAD_ARG, "metric wsum needs metric_w");
    if (!(eps > 0.0) || !(1.0 / eps < 1e300)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    return 0;
}

// one launch over the batches [i0, i0 + cnt) (cnt <= SYMPA_MAX_FUSED_BATCHES)
int launch_group(PackedArgs& a, int n, int model, const int64_t* const* idx1, const int64_t* const* idx2, const int64_t* b,
                 double* const* out, int cnt, hipStream_t s) {
    uint64_t tiles = 0;
    int k = 0;
    for (int i = 0; i < cnt; ++i) {
        if (b[i] == 0) continue;
        tiles += (uint64_t)((b[i] + 63) / 64);
        a.idx1[k] = idx1[i];
        a.idx2[k] = idx2[i];
        a.out[k] = out[i];
        a.b[k] = b[i];
        a.tile_end[k] = (unsigned)tiles;
        ++k;
    }
    if (k == 0) return 0;
    for (int i = k; i < SYMPA_MAX_FUSED_BATCHES; ++i) {
        a.tile_end[i] = (unsigned)tiles;
        a.b[i] = 1;
        a.idx1[i] = a.idx1[0];
        a.idx2[i] = a.idx2[0];
        a.out[i] = a.out[0];
    }
    a.num_batches = k;
    a.tiles = (unsigned)tiles;
    const unsigned grid = (unsigned)cu_count();          // (the launcher turns it into resident one-wave blocks)
    // staggered first round (siegel_dist_kernel.hpp): dims 7, 8, upper model, tables beyond the L2s, two rounds or more
    a.stagger = (n >= 7 && model == SYMPA_MODEL_UPPER && a.tiles >= 8 * grid &&
                 a.num_rows * (int64_t)pack_row_doubles(n, model) * 8 >= ((int64_t)12 << 20)) ? 1 : 0;
    switch (n) {
        case 5: return forward_n<5>(a, grid, model, s);
        case 6: return forward_n<6>(a, grid, model, s);
        case 7: return forward_n<7>(a, grid, model, s);
        case 8: return forward_n<8>(a, grid, model, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "packed forward: dims 5..8");
}

}  // namespace

namespace sympa_hip {

int launch_dense_persistent(PackedArgs& a, int n, hipStream_t s) {
    uint64_t tiles = 0;
    for (int i = 0; i < a.num_batches; ++i) {
        tiles += (uint64_t)((a.b[i] + 63) / 64);
        if (tiles > 0x7fffffffull) return fail(SYMPA_ERR_BAD_ARG, "batches too large for one launch");
        a.tile_end[i] = (unsigned)tiles;
    }
    if (tiles == 0) return 0;
    for (int i = a.num_batches; i < SYMPA_MAX_FUSED_BATCHES; ++i) {
        a.tile_end[i] = (unsigned)tiles;
        a.b[i] = 1;
        a.idx1[i] = a.idx1[0];
        a.idx2[i] = a.idx2[0];
        a.out[i] = a.out[0];
    }
    a.tiles = (unsigned)tiles;
    const unsigned cus = (unsigned)cu_count();
    a.stagger = (n >= 7 && !a.identity && a.tiles >= 8 * cus && a.num_rows * (int64_t)(16 * n * n) >= ((int64_t)12 << 20)) ? 1 : 0;
    switch (n) {
        case 7: return launch_dense_forward<7>(a, cus, s);
        case 8: return launch_dense_forward<8>(a, cus, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "persistent dense forward: dims 7, 8");
}

}  // namespace sympa_hip

extern "C" {

int64_t sympa_table_pack_bytes(int64_t num_rows, int n, int model) {
    if (num_rows <= 0 || !packed_dims_ok(n)) return 0;
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return 0;
    return num_rows * (int64_t)pack_row_doubles(n, model) * 8;
}

int sympa_table_pack(const double* table, int64_t num_rows, int n, int model, void* pack, int64_t pack_bytes, int32_t* status,
                     void* stream) {
    return pack_any(table, num_rows, n, model, pack, pack_bytes, status, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int sympa_table_pack_refresh(const double* table, int64_t num_rows, int n, int model, void* pack, int64_t pack_bytes,
                             void* digest_state, int flags, int32_t* status, void* stream) {
    if (!packed_dims_ok(n)) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "packed table: dims 5..8");
    if (table == nullptr || num_rows <= 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (digest_state == nullptr) return fail(SYMPA_ERR_BAD_ARG, "pack refresh: null digest state");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rc = launch_table_digest(table, num_rows * (int64_t)(16 * n * n), digest_state, (flags & SYMPA_FLAG_DIGEST_FORCE) ? 1 : 0, s);
    if (rc != 0) return rc;
    return pack_any(table, num_rows, n, model, pack, pack_bytes, status,
                    reinterpret_cast<const unsigned*>(digest_state) + DIGEST_GUARD_WORD, s);
}

int sympa_model_forward_packed(const void* pack, int64_t pack_bytes, int64_t num_rows, int n, const int64_t* src,
                               int64_t src_stride, const int64_t* dst, int64_t dst_stride, int64_t b, int model, int metric,
                               const double* metric_w, double eps, const double* scale, double scale_coef, double* out,
                               int32_t* status, int flags, void* stream) {
    (void)flags;
    if (b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
    if (b == 0) return 0;
    const int rc = check_common(pack, pack_bytes, num_rows, n, model, metric, metric_w, eps, scale, scale_coef);
    if (rc != 0) return rc;
    if (src == nullptr || dst == nullptr || out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (b > (int64_t)0x7fffffff * 32) return fail(SYMPA_ERR_BAD_ARG, "batch too large for one launch");
    PackedArgs a;
    std::memset(&a, 0, sizeof(a));
    a.pack = reinterpret_cast<const double*>(pack);
    a.base2 = a.pack;
    a.num_rows = num_rows;
    a.stride1 = src_stride;
    a.stride2 = dst_stride;
    a.metric_w = metric_w;
    a.scale = scale;
    a.inv_scale_coef = 1.0 / scale_coef;
    a.inv_eps = 1.0 / eps;
    a.status = status;
    a.metric = metric;
    return launch_group(a, n, model, &src, &dst, &b, &out, 1, reinterpret_cast<hipStream_t>(stream));
}

int sympa_model_forward_batches_packed(const void* pack, int64_t pack_bytes, int64_t num_rows, int n,
                                       const int64_t* const* triplets, int64_t stride, const int64_t* b, int num_batches,
                                       int model, int metric, const double* metric_w, double eps, const double* scale,
                                       double scale_coef, double* const* out, int32_t* status, int flags, void* stream) {
    (void)flags;
    if (num_batches < 0) return fail(SYMPA_ERR_BAD_ARG, "bad batch list");
    if (num_batches == 0) return 0;
    if (triplets == nullptr || b == nullptr || out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null batch list");
    if (stride < 2) return fail(SYMPA_ERR_BAD_ARG, "triplet stride must be >= 2");
    const int rc = check_common(pack, pack_bytes, num_rows, n, model, metric, metric_w, eps, scale, scale_coef);
    if (rc != 0) return rc;
    PackedArgs a;
    std::memset(&a, 0, sizeof(a));
    a.pack = reinterpret_cast<const double*>(pack);
    a.base2 = a.pack;
    a.num_rows = num_rows;
    a.stride1 = stride;
    a.stride2 = stride;
    a.metric_w = metric_w;
    a.scale = scale;
    a.inv_scale_coef = 1.0 / scale_coef;
    a.inv_eps = 1.0 / eps;
    a.status = status;
    a.metric = metric;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // groups of up to SYMPA_MAX_FUSED_BATCHES consecutive batches per launch
    const int64_t* idx1[SYMPA_MAX_FUSED_BATCHES];
    const int64_t* idx2[SYMPA_MAX_FUSED_BATCHES];
    int i0 = 0;
    while (i0 < num_batches) {
        int cnt = 0;
        int64_t tiles = 0;
        while (i0 + cnt < num_batches && cnt < SYMPA_MAX_FUSED_BATCHES) {
            const int64_t bi = b[i0 + cnt];
            if (bi < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
            if (bi > 0 && (triplets[i0 + cnt] == nullptr || out[i0 + cnt] == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
            const int64_t t = (bi + 63) / 64;
            if (tiles + t > (int64_t)0x7fffffff) break;
            tiles += t;
            idx1[cnt] = triplets[i0 + cnt];
            idx2[cnt] = triplets[i0 + cnt] + 1;
            ++cnt;
        }
        if (cnt == 0) return fail(SYMPA_ERR_BAD_ARG, "batch too large for one launch");
        const int rc2 = launch_group(a, n, model, idx1, idx2, b + i0, out + i0, cnt, s);
        if (rc2 != 0) return rc2;
        i0 += cnt;
    }
    return 0;
}

}  // extern "C"
