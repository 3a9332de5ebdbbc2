// gfx950 backward kernels + C-ABI (sympa_siegel_dist_bwd, sympa_model_backward, sympa_model_loss_backward).
#include "siegel_coop_bwd_kernel.hpp"
#include "siegel_bwd_split_kernel.hpp"

namespace sympa_hip {
int launch_bwd_one_lane(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s) {
    switch (n) {
        case 1: return launch_bwd_n<1>(a, model, scatter, s);
        case 2: return launch_bwd_n<2>(a, model, scatter, s);
        case 3: return launch_bwd_n<3>(a, model, scatter, s);
        case 4: return launch_bwd_n<4>(a, model, scatter, s);
        case 5: return launch_bwd_n<5>(a, model, scatter, s);
        case 6: return launch_bwd_n<6>(a, model, scatter, s);
        case 7: return model == SYMPA_MODEL_UPPER ? launch_bwd_n7_upper(a, scatter, s) : launch_bwd_n7_bounded(a, scatter, s);
        case 8: return model == SYMPA_MODEL_UPPER ? launch_bwd_n8_upper(a, scatter, s) : launch_bwd_n8_bounded(a, scatter, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "one-pair-per-lane backward: dims 1..8");
}
}  // namespace sympa_hip

namespace {
using namespace sympa_hip;

int launch_bwd(const BwdArgs& a, int n, int model, bool scatter, void* workspace, int64_t workspace_bytes, void* stream) {
    const int rc = validate(a.f, model, n);
    if (rc != 0) return rc;
    if (a.f.b == 0) return 0;
    if ((a.go == nullptr && a.graph_dist == nullptr) || a.g1 == nullptr || a.g2 == nullptr)
        return fail(SYMPA_ERR_BAD_ARG, "null gradient buffer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // SYMPA_FLAG_MERGE_SRC with per-pair rows: only the one-lane kernels of dims <= 6 leave the merged layout (one row per run of equal
    // source ids); any other kernel would write every row while the caller's slot list (ops.sorted_slots(merged_src=b)) drops all but
    // the run ends -- a silently wrong gradient (round-5 advice).  The scatter forms ignore the flag harmlessly (every row is added).
    const bool merged_rows = (a.f.flags & SYMPA_FLAG_MERGE_SRC) != 0 && !scatter;
    const auto no_merge = []() {
        return fail(SYMPA_ERR_BAD_ARG, "SYMPA_FLAG_MERGE_SRC with grad_rows: only the one-pair-per-lane kernels of dims <= 6 write merged "
                                       "rows (no workspace / SYMPA_FLAG_SPLIT / SYMPA_FLAG_COOP / SYMPA_FLAG_GENERIC, dims <= 6)");
    };
    if (merged_rows && (n > 6 || (a.f.flags & (SYMPA_FLAG_COOP | SYMPA_FLAG_SPLIT)))) return no_merge();
    // Eight lanes per pair (two pairs per DPP row, no scratch) where measured faster than one pair per lane
    // (tools/bwd_coop_ab_small.py, per 262 144 pairs): the fused step at n = 8 (upper 1.78 -> 1.49 ms, bounded 2.59 -> 1.99 ms),
    // bounded n = 8 dense rows (2.51 -> 2.15 ms), bounded n = 7 fused (1.70 -> 1.56 ms).  SYMPA_FLAG_COOP forces it for
    // dims 5..8, SYMPA_FLAG_GENERIC forces the one-pair-per-lane kernels.
    // the deterministic per-wave sums exist in the one-pair-per-lane kernels only (dims <= 8); the training graph's batch
    // window (step_counter) in every kernel family but the rolled one-lane kernels of dims 9..16
    // dims 5..8 with a workspace: the split backward (siegel_bwd_split_kernel.hpp), ONE pair per lane in two kernels -- the
    // eigen-decomposition with vectors is no longer redundant in the lanes of a pair (fused step, upper n = 8, 262 144 pairs:
    // 1.42 ms eight lanes per pair -> 0.81 ms (0.68 ms on batches sorted by source row), profiles/r04_n8_backward_split.txt).  SYMPA_FLAG_GENERIC / SYMPA_FLAG_COOP or no
    // (or too small a) workspace: the kernels below, as before.
    // Default where measured faster (tools/bwd_split_ab.py, fused step): upper n = 7 200 against 266 us (one pair per lane, one
    // kernel) per 65 536 pairs, n = 8 811 against 1 421 us (eight lanes per pair) per 262 144; dims 5, 6 fit one lane's registers
    // in one kernel (81 / 116 against 115 / 141 us) and the bounded model's second stage still spills (n = 8: 2.83 against 2.05 ms).
    if (n >= 5 && n <= 8 && workspace != nullptr && !(a.f.flags & (SYMPA_FLAG_GENERIC | SYMPA_FLAG_COOP)) &&
        ((model == SYMPA_MODEL_UPPER && n >= 7) || (a.f.flags & SYMPA_FLAG_SPLIT)) &&
        bwd_split_available(n, model) && workspace_bytes >= bwd_split_workspace_bytes(a.f.b, n, model) &&
        a.f.b < ((int64_t)1 << 29) - 64)          // its kernels address a pair inside a workspace entry with a 32-bit byte offset
        return launch_bwd_split(a, n, model, scatter, workspace, workspace_bytes, s);
    const bool det_mode = a.wave_partials != nullptr;
    if (det_mode && n > 8) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "wave_partials: dims 1..8");
    const bool one_lane = (a.f.flags & SYMPA_FLAG_GENERIC) || instance_fallback(SYMPA_FAMILY_SIEGEL_BWD, model, n) || det_mode;
    if (a.f.batch_counter != nullptr && n > 8 && one_lane)
        return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "step_counter at dims 9..16: the sixteen-lanes kernels only");
    if (n >= 5 && n <= 8 && !one_lane) {
        const bool bounded = model == SYMPA_MODEL_BOUNDED;
        const bool faster = (n == 8 && (scatter || bounded)) || (n == 7 && bounded && scatter);
        if (faster || (a.f.flags & SYMPA_FLAG_COOP)) return launch_bwd_half(a, n, model, scatter, s);
    }
    if (n >= 1 && n <= 8) return launch_bwd_one_lane(a, n, model, scatter, s);
    if (n > 8 && n <= SYMPA_MAX_DIMS_BACKWARD) {
        // sixteen lanes per pair (siegel_coop_bwd.hpp); SYMPA_FLAG_GENERIC keeps the one-lane-per-pair kernel over scratch
        if (!one_lane) return launch_bwd_coop(a, n, model, scatter, s);
        return launch_bwd_rolled(a, n, model, scatter, s);
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "backward: dims outside [1, SYMPA_MAX_DIMS_BACKWARD]");
}

}  // namespace

extern "C" {

int64_t sympa_siegel_backward_workspace_bytes(int64_t b, int n, int model) { return bwd_split_workspace_bytes(b, n, model); }

int sympa_siegel_dist_bwd(const double* z1, const double* z2, const double* grad_out, int64_t b, int n, int model,
                          int metric, const double* metric_w, double eps, double* grad_z1, double* grad_z2,
                          double* grad_w, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream) {
    BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.f.base1 = z1;
    a.f.base2 = z2;
    a.f.b = b;
    a.f.num_rows = b;
    a.f.metric_w = metric_w;
    a.f.inv_scale_coef = 1.0;
    a.f.inv_eps = 1.0 / eps;
    a.f.status = status;
    a.f.metric = metric;
    a.f.flags = flags;
    a.go = grad_out;
    a.g1 = grad_z1;
    a.g2 = grad_z2;
    a.gw = grad_w;
    return launch_bwd(a, n, model, false, workspace, workspace_bytes, stream);
}

int sympa_model_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                         const int64_t* dst, int64_t dst_stride, int64_t b, int model, int metric,
                         const double* metric_w, double eps, const double* scale, double scale_coef,
                         const double* grad_out, double* grad_table, double* grad_w, double* grad_scale,
                         double* out, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream) {
    if (b > 0 && (src == nullptr || dst == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null index buffer");
    if (num_rows <= 0 && b > 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.f.base1 = table;
    a.f.base2 = table;
    a.f.idx1 = src;
    a.f.idx2 = dst;
    a.f.idx1_stride = src_stride;
    a.f.idx2_stride = dst_stride;
    a.f.num_rows = num_rows;
    a.f.b = b;
    a.f.metric_w = metric_w;
    a.f.scale = scale;
    a.f.inv_scale_coef = 1.0 / scale_coef;
    a.f.inv_eps = 1.0 / eps;
    a.f.out = out;
    a.f.status = status;
    a.f.metric = metric;
    a.f.flags = flags;
    a.go = grad_out;
    a.g1 = grad_table;
    a.g2 = grad_table;
    a.gw = grad_w;
    a.gscale = grad_scale;
    return launch_bwd(a, n, model, true, workspace, workspace_bytes, stream);
}

int sympa_model_loss_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                              const int64_t* dst, int64_t dst_stride, const double* graph_dist, int64_t b, int model,
                              int metric, const double* metric_w, double eps, const double* scale, double scale_coef,
                              double loss_scale, double* loss, double* grad_table, double* grad_w, double* grad_scale,
                              double* out, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream) {
    if (b > 0 && (src == nullptr || dst == nullptr || graph_dist == nullptr))
        return fail(SYMPA_ERR_BAD_ARG, "null index / graph-distance buffer");
    if (num_rows <= 0 && b > 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.f.base1 = table;
    a.f.base2 = table;
    a.f.idx1 = src;
    a.f.idx2 = dst;
    a.f.idx1_stride = src_stride;
    a.f.idx2_stride = dst_stride;
    a.f.num_rows = num_rows;
    a.f.b = b;
    a.f.metric_w = metric_w;
    a.f.scale = scale;
    a.f.inv_scale_coef = 1.0 / scale_coef;
    a.f.inv_eps = 1.0 / eps;
    a.f.out = out;
    a.f.status = status;
    a.f.metric = metric;
    a.f.flags = flags;
    a.g1 = grad_table;
    a.g2 = grad_table;
    a.gw = grad_w;
    a.gscale = grad_scale;
    a.graph_dist = graph_dist;
    a.loss = loss;
    a.loss_scale = loss_scale;
    return launch_bwd(a, n, model, true, workspace, workspace_bytes, stream);
}

int sympa_model_train_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                               const int64_t* dst, int64_t dst_stride, const double* graph_dist, int64_t b,
                               const int64_t* step_counter, int model, int metric, const double* metric_w, double eps,
                               const double* scale, double scale_coef, double loss_scale, double* loss, double* grad_table,
                               double* grad_rows, double* grad_w, double* grad_scale, double* wave_partials, int32_t* status,
                               void* workspace, int64_t workspace_bytes, int flags, void* stream) {
    if (b > 0 && (src == nullptr || dst == nullptr || graph_dist == nullptr))
        return fail(SYMPA_ERR_BAD_ARG, "null index / graph-distance buffer");
    if (num_rows <= 0 && b > 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    if ((grad_table == nullptr) == (grad_rows == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "give grad_table or grad_rows, not both");
    if (wave_partials != nullptr && grad_rows == nullptr)
        return fail(SYMPA_ERR_BAD_ARG, "wave_partials belongs to the rows form (the scatter form is atomic anyway)");
    BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.f.base1 = table;
    a.f.base2 = table;
    a.f.idx1 = src;
    a.f.idx2 = dst;
    a.f.idx1_stride = src_stride;
    a.f.idx2_stride = dst_stride;
    a.f.num_rows = num_rows;
    a.f.b = b;
    a.f.metric_w = metric_w;
    a.f.scale = scale;
    a.f.inv_scale_coef = 1.0 / scale_coef;
    a.f.inv_eps = 1.0 / eps;
    a.f.status = status;
    a.f.metric = metric;
    a.f.flags = flags;
    a.f.batch_counter = step_counter;
    a.gw = grad_w;
    a.gscale = grad_scale;
    a.graph_dist = graph_dist;
    a.loss = loss;
    a.loss_scale = loss_scale;
    a.wave_partials = wave_partials;
    if (grad_table != nullptr) {
        a.g1 = grad_table;
        a.g2 = grad_table;
        return launch_bwd(a, n, model, true, workspace, workspace_bytes, stream);
    }
    a.g1 = grad_rows;
    a.g2 = grad_rows + b * 2 * (int64_t)n * n;
    return launch_bwd(a, n, model, false, workspace, workspace_bytes, stream);
}

int sympa_model_loss_backward_rows(const double* table, int64_t num_rows, int n, const int64_t* src,
                                   int64_t src_stride, const int64_t* dst, int64_t dst_stride, const double* graph_dist,
                                   int64_t b, int model, int metric, const double* metric_w, double eps,
                                   const double* scale, double scale_coef, double loss_scale, double* loss,
                                   double* grad_src_rows, double* grad_dst_rows, double* grad_w, double* grad_scale,
                                   double* out, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream) {
    if (b > 0 && (src == nullptr || dst == nullptr || graph_dist == nullptr))
        return fail(SYMPA_ERR_BAD_ARG, "null index / graph-distance buffer");
    if (num_rows <= 0 && b > 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.f.base1 = table;
    a.f.base2 = table;
    a.f.idx1 = src;
    a.f.idx2 = dst;
    a.f.idx1_stride = src_stride;
    a.f.idx2_stride = dst_stride;
    a.f.num_rows = num_rows;
    a.f.b = b;
    a.f.metric_w = metric_w;
    a.f.scale = scale;
    a.f.inv_scale_coef = 1.0 / scale_coef;
    a.f.inv_eps = 1.0 / eps;
    a.f.out = out;
    a.f.status = status;
    a.f.metric = metric;
    a.f.flags = flags;
    a.g1 = grad_src_rows;
    a.g2 = grad_dst_rows;
    a.gw = grad_w;
    a.gscale = grad_scale;
    a.graph_dist = graph_dist;
    a.loss = loss;
    a.loss_scale = loss_scale;
    return launch_bwd(a, n, model, false, workspace, workspace_bytes, stream);
}

}  // extern "C"
