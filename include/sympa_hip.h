/* sympa_hip.h -- C-ABI of the MI355X (gfx950) Siegel-distance library  (libsympa_hip.so)
 *
 * The reference (fedelopez77/sympa) is pure Python/torch and has no FFI layer; its drop-in boundary
 * for this path is the Python API  Model.forward -> manifold.dist  (SURVEY.md section 8b).  These
 * entry points are what a binding for that path binds: one call per reference method, plain
 * pointers and sizes, no torch types.  Each entry cites the reference interface it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated otherwise; the caller owns all buffers,
 *     the library allocates nothing and keeps no mutable global state (re-entrant);
 *   - a point is [2, n, n] fp64 row-major, plane 0 = Re, plane 1 = Im (sympa/math/csym_math.py:1-8);
 *     only the upper triangle (i <= j) of each plane is read (points on the manifold are symmetric);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream) and the
 *     call returns without synchronising; it is safe to capture into a hipGraph;
 *   - return value: 0 = enqueued, <0 = argument error (SYMPA_ERR_*), >0 = hipError_t of the launch;
 *     sympa_last_error() gives a thread-local message;
 *   - numeric-range violations the reference reports with `assert` (siegel_manifold.py:64-66) or a
 *     Python IndexError are reported through `status` (device int32[2], may be NULL):
 *         status[0] |= SYMPA_ST_* bits,   status[1] += number of offending pairs.
 *     The caller zeroes it and reads it back lazily (no host sync inside the library).
 */
#ifndef SYMPA_HIP_H
#define SYMPA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* models: ManifoldFactory.sympa_manifolds keys (sympa/embeddings.py:145-149) */
#define SYMPA_MODEL_UPPER 0   /* UpperHalfManifold   sympa/manifolds/upper_half.py:8 */
#define SYMPA_MODEL_BOUNDED 1 /* BoundedDomainManifold sympa/manifolds/bounded_domain.py:10 */

/* metrics: MetricType values (sympa/manifolds/metrics.py:6-12) */
#define SYMPA_METRIC_RIEM 0
#define SYMPA_METRIC_FONE 1
#define SYMPA_METRIC_FINF 2
#define SYMPA_METRIC_FMIN 3
#define SYMPA_METRIC_WSUM 4

#define SYMPA_ST_NOT_PD 1         /* Im(z) (upper) or I - z z^H (bounded) not positive definite */
#define SYMPA_ST_NONFINITE 2      /* result is NaN/Inf */
#define SYMPA_ST_BAD_INDEX 4      /* gather index outside [0, num_rows) (reference: IndexError) */
#define SYMPA_ST_NO_CONVERGENCE 8 /* eigenvalue iteration hit its sweep cap */

#define SYMPA_ERR_BAD_ARG (-1)
#define SYMPA_ERR_UNSUPPORTED_DIMS (-2)

/* flags */
#define SYMPA_FLAG_LOW_LDS 1 /* stage one endpoint at a time: half the LDS per block, so blocks of two launches
                                that overlap (different streams / independent graph nodes) share a CU */

#define SYMPA_FLAG_GENERIC 2 /* run the runtime-n one-lane-per-pair kernel even where a sixteen-lanes-per-pair one applies
                                (spd n >= 6, upper / bounded n >= 9); the tests cross-check the two */

#define SYMPA_FLAG_ANY_ORDER 4 /* forward launches only: dispatch without the in-order barrier bit (hipExtAnyOrderLaunch), so
                                  the launch may start while earlier launches of the same stream still run.  Only for
                                  launches that do not depend on earlier work of that stream (independent batches);
                                  a later ordinary launch or synchronisation still waits for all of them.  MEASURED on
                                  gfx950 / ROCm 7.2 (profiles/r02_launch_overhead.txt): no overlap is obtained -- the
                                  runtime documents the flag as unsupported on GFX9 -- so the fused form below
                                  (SYMPA_FLAG_FUSE) is what overlaps independent batches */

#define SYMPA_FLAG_FUSE 8 /* sympa_model_forward_batches only (dims <= SYMPA_MAX_DIMS): up to SYMPA_MAX_FUSED_BATCHES
                             consecutive batches per kernel launch instead of one launch per batch */
#define SYMPA_MAX_FUSED_BATCHES 32
#define SYMPA_FLAG_COOP 32 /* A/B switch of the sixteen-lanes-per-pair layout (DESIGN.md sections 5, 8, 10):
                              forward, dims 6 and 8: run it instead of the one-pair-per-lane kernel (2-2.5x slower there);
                              backward entries, dims 5..8: the eight-lanes-per-pair kernels (two pairs per DPP row; the default
                              where measured faster: the fused step at n = 8, bounded n = 8 rows, bounded n = 7 fused);
                              SYMPA_FLAG_GENERIC forces the one-pair-per-lane kernels there;
                              sympa_spd_backward_rows: the single-round kernel instead of the one that runs the QL of two
                              rounds together */
#define SYMPA_FLAG_SPLIT 64 /* backward entries, dims 5..8, with a workspace: run the split (two-kernel, one pair per lane) backward
                             * wherever it is built -- by default it runs only where it measured faster than the one-launch kernels
                             * (upper model, dims 7 and 8: profiles/r04_n8_backward_split.txt) */
#define SYMPA_FLAG_NO_SYMMETRY 16 /* sympa_all_pairs_dist_packed: evaluate both (i, j) and (j, i) even for the full matrix */
#define SYMPA_FLAG_MERGE_SRC 128 /* sympa_model_train_backward, dims <= 6: consecutive pairs of a wave (64 pairs) with the SAME source id
                                  * are summed in pair order inside the wave's LDS tile.  Rows form (grad_rows): the sum is written ONCE,
                                  * into the source row slot of the LAST pair of the run, the other source slots of the run are NOT
                                  * written -- the caller's segmented sum must leave them out (sympa_amd.ops.sorted_slots(...,
                                  * merged_src=batch)); bitwise reproducible like the plain rows form.  Scatter form (grad_table): one
                                  * atomic row per run.  For batches sorted by their first column (~13 pairs per source row at the
                                  * headline shape) that is 45 % fewer rows through memory. */

#define SYMPA_MAX_DIMS 8          /* largest n with a register-resident forward kernel in this build */
#define SYMPA_MAX_DIMS_BACKWARD 16 /* largest n with a backward kernel in this build: n <= 6 one pair per lane in registers;
                                      n = 7, 8 one pair per lane or eight lanes per pair (the default where measured faster;
                                      SYMPA_FLAG_COOP forces the eight-lanes kernels at n = 5..8, SYMPA_FLAG_GENERIC the
                                      one-pair-per-lane ones); n = 9..16 sixteen lanes per pair (csrc/siegel_coop_bwd.hpp),
                                      SYMPA_FLAG_GENERIC selects the same adjoint as rolled loops over per-lane scratch */
#define SYMPA_MAX_DIMS_ALL_PAIRS_PACKED 8 /* sympa_all_pairs_dist_packed: per-point factor reuse, dims 1..8 (dims 8: the 64
                                             packed column points of a block live in an LDS tile, 55 / 70 KB) */
#define SYMPA_MAX_DIMS_GENERIC 16 /* n in (SYMPA_MAX_DIMS, 16]: the forward runs sixteen lanes per pair (csrc/siegel_coop.hpp;
                                     SYMPA_FLAG_GENERIC selects the runtime-n fallback over scratch).  The table operations
                                     (sympa_egrad2rgrad / sympa_projx / sympa_rsgd_step* / sympa_tangent_sqnorm) run one row
                                     per lane for n <= 6, eight lanes per row at n = 7, 8 and sixteen at n = 9..16
                                     (csrc/siegel_coop_table.hpp); the rolled one-row-per-lane kernels over scratch
                                     (siegel_table_rolled.hip) remain for the gated exact projection and behind
                                     SYMPA_TABLE_GENERIC=1 / an instance fallback */

/* Library / build identification. */
const char* sympa_version(void);
const char* sympa_last_error(void);
int sympa_max_dims(void);

/* Kernel families whose default instantiations use the sixteen- / eight-lanes-per-pair layout with inline-asm DPP
 * instructions (the compiler cannot see their hazards; the build scans the ISA, and a numerical self-check
 * compares every instantiation with the one-lane kernel on first use -- sympa_amd/selfcheck.py).  An instantiation
 * (family, model, n) marked here is routed to the one-lane-per-pair / one-row-per-lane kernel by every entry point, whatever
 * the flags; model is SYMPA_MODEL_* (0 for the spd families).  Process-wide, thread-safe, no GPU call.  No reference
 * counterpart (the reference has no native kernels). */
#define SYMPA_FAMILY_SIEGEL_FWD 0   /* sympa_siegel_dist_fwd / sympa_model_forward*, n = 9..16 */
#define SYMPA_FAMILY_SIEGEL_BWD 1   /* sympa_siegel_dist_bwd / sympa_model_backward / sympa_model_loss_backward*, n = 5..16 */
#define SYMPA_FAMILY_SIEGEL_TABLE 2 /* sympa_egrad2rgrad / sympa_projx / sympa_rsgd_step* / sympa_tangent_sqnorm, n = 7..16 */
#define SYMPA_FAMILY_SPD_FWD 3      /* sympa_spd_dist_fwd / sympa_spd_model_forward, n = 6..16 */
#define SYMPA_FAMILY_SPD_BWD 4      /* sympa_spd_backward_rows / sympa_spd_loss_backward, n = 3..16 */
#define SYMPA_FAMILY_SPD_TABLE 5    /* sympa_spd_egrad2rgrad / sympa_spd_rsgd_step, n = 3..16 */
#define SYMPA_NUM_FAMILIES 6
int sympa_set_instance_fallback(int family, int model, int n, int on);
int sympa_get_instance_fallback(int family, int model, int n);

/* manifold.dist(z1, z2) for pre-gathered points.
 * Replaces SiegelManifold.dist (sympa/manifolds/siegel_manifold.py:41-72) for model = UPPER and
 * BoundedDomainManifold.dist (sympa/manifolds/bounded_domain.py:27-39) for model = BOUNDED,
 * including Metric.compute_metric (sympa/manifolds/metrics.py:42-121).
 *   z1, z2    [b, 2, n, n] fp64 contiguous
 *   metric_w  [n] fp64, the wsum weights (metrics.py:103-108); may be NULL unless metric = WSUM
 *   eps       clamp of (1 - d), reference EPS[float64] = 1e-5 (sympa/config.py:19)
 *   out       [b] fp64 distances
 *   vvd_out   [b, n] fp64 ascending vector-valued distance v (siegel_manifold.py:69-70), or NULL
 *   flags     0 or SYMPA_FLAG_*
 */
int sympa_siegel_dist_fwd(const double* z1, const double* z2, int64_t b, int n, int model, int metric,
                          const double* metric_w, double eps, double* out, double* vvd_out, int32_t* status,
                          int flags, void* stream);

/* Model.forward(input_triplet) fused: gather two table rows per pair, distance, times the scale.
 * Replaces Model.forward / Model.distance / Model.get_scale (sympa/model.py:16-41) and
 * Embeddings.forward (sympa/embeddings.py:29-34).
 *   table      [num_rows, 2, n, n] fp64 (the ManifoldParameter, embeddings.py:27,68)
 *   src, dst   int64 node ids, element i at src[i * src_stride] (so input_triplet[:, 0] / [:, 1] of a
 *              [b, 2|3] int64 tensor are passed without a copy: stride 2 or 3)
 *   scale      device pointer to the model's scale parameter (1 fp64) or NULL for 1.0;
 *              out = dist * max(scale / scale_coef, 0.1)          (model.py:40-41)
 */
int sympa_model_forward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                        const int64_t* dst, int64_t dst_stride, int64_t b, int model, int metric,
                        const double* metric_w, double eps, const double* scale, double scale_coef, double* out,
                        int32_t* status, int flags, void* stream);

/* Model.forward over a LIST of batches: what Runner.evaluate's loop (sympa/runner.py:126-137) and the batch loop of
 * Runner.train_epoch's forward (runner.py:98-103) issue one call at a time.  Batch i reads the int64 triplets
 * triplets[i] ([b[i], stride], src = column 0, dst = column 1) and writes out[i] ([b[i]] fp64); `triplets`, `b`,
 * `out` and `streams` are HOST arrays (of device pointers / sizes / hipStream_t).  Launch i goes to
 * streams[i % num_streams]: the batches are independent, so consecutive launches overlap on the GPU (the next batch
 * gathers its rows while the previous one computes) -- with SYMPA_FLAG_LOW_LDS three blocks of different launches share
 * a CU.  One kernel launch per batch, exactly as num_batches calls of sympa_model_forward; nothing is synchronised.
 * With SYMPA_FLAG_FUSE (dims <= SYMPA_MAX_DIMS) groups of up to SYMPA_MAX_FUSED_BATCHES consecutive batches are
 * evaluated by ONE launch each (group g on streams[g % num_streams]): the grid is several blocks per CU deep, so
 * the batches overlap inside the launch with no kernel boundary between them -- the form to use when a batch is
 * about one wave per SIMD or less (65 536 pairs fill the 1 024 SIMDs of an MI355X exactly once).  Same results,
 * bit for bit. */
int sympa_model_forward_batches(const double* table, int64_t num_rows, int n, const int64_t* const* triplets,
                                int64_t stride, const int64_t* b, int num_batches, int model, int metric,
                                const double* metric_w, double eps, const double* scale, double scale_coef,
                                double* const* out, int32_t* status, int flags, void* const* streams, int num_streams);

/* ---- Packed table for the INDEXED forward, dims 5..8 (csrc/siegel_packed_kernel.hpp) ----------------------------------
 * The reference's loops run many batches over an unchanged table: Runner.evaluate (sympa/runner.py:124-135), the N forward
 * calls of the mAP matrix (runner.py:142-154), every Model.forward between two optimiser steps (sympa/model.py:16-30 with
 * Embeddings.forward, sympa/embeddings.py:29-34).  sympa_table_pack factors every point ONCE per table version -- Y = L L^T
 * (upper) / I - W W^H = C C^H (bounded) -- and writes one contiguous row per point: the upper triangles of both planes and the
 * INVERTED factor (row stride sympa_table_pack_bytes(1, n, model) bytes: 864 at n = 8 upper, where the reference row is 1 024).
 * The pair kernels then need no Cholesky and no solve: E = A1 (Z2 - Z1) A2^T.  A point outside the manifold is reported through
 * `status` here (SYMPA_ST_NOT_PD; `status` may be NULL) and flags every pair it enters later.  (sympa_amd passes NULL: the
 * reference's assertions sit in dist, siegel_manifold.py:64-66, so a point no batch touches raises nothing there.)
 *   pack   caller-owned device buffer, 16-byte aligned, sympa_table_pack_bytes(num_rows, n, model) bytes (0 = this build has no
 *          packed path for these dims / model: use sympa_model_forward); valid until the table changes (the caller keeps the
 *          version: sympa_amd/model.py repacks when the ManifoldParameter's version counter has moved).
 * sympa_model_forward_packed = sympa_model_forward reading `pack` instead of the table: one pair per lane, PERSISTENT one-wave blocks
 * (grid = what the chip holds at once), the next tile's ids and first rows in flight behind the current tile's arithmetic, the first
 * point's triangles subtracted from the second's in place as they arrive (no scratch in any instantiation: bounded n = 8 runs 245 ->
 * 163 us per 262 144 pairs, upper n = 8 135 -> 127, n = 6 85 -> 68; profiles/r05_packed_forward.txt).  Same distances as
 * sympa_model_forward to rounding (~1e-15 relative on the bench tables; the arithmetic differs: products with the inverted factor
 * instead of triangular solves).
 * sympa_model_forward_batches_packed: the list form (sympa_model_forward_batches); up to SYMPA_MAX_FUSED_BATCHES consecutive batches
 * share a launch; one stream. */
int64_t sympa_table_pack_bytes(int64_t num_rows, int n, int model);
int sympa_table_pack(const double* table, int64_t num_rows, int n, int model, void* pack, int64_t pack_bytes, int32_t* status,
                     void* stream);

/* Validity of a pack decided ON THE DEVICE (csrc/table_digest.hpp).  The reference writes its table through `.data`
 * (sympa/embeddings.py:36-39 `self.embeds.data = self.manifold.projx(...)`; geoopt / torch-1.5 optimisers `p.data.add_(...)`), which
 * no version counter sees, so a drop-in must not trust one.
 * sympa_table_digest: one kernel reads `bytes` bytes at `data` (16-byte aligned, a multiple of 8) and sums a 64-bit position-
 * weighted digest; its last block compares it with the one kept in `state` (caller-owned, SYMPA_DIGEST_STATE_BYTES of device memory,
 * zeroed once by the caller), stores the new one and writes ((uint32_t*)state)[SYMPA_DIGEST_CHANGED_WORD] = 1 when the bytes
 * changed since the previous call on this state (or with SYMPA_FLAG_DIGEST_FORCE), else 0.  No host synchronisation.
 * sympa_table_pack_refresh = that digest of the table + sympa_table_pack whose blocks return at once when the word is 0: the pack
 * is remade exactly when the table's bytes differ from those it was made from (first call on a zeroed state: always), on the
 * stream, graph-capturable -- a replayed graph with an optimiser step in front repacks by itself.  Cost when nothing changed: one
 * read of the table (measured: profiles/r06_pack_refresh.txt).  sympa_spd_table_pack_refresh: the same for sympa_spd_table_pack. */
#define SYMPA_DIGEST_STATE_BYTES 4096
#define SYMPA_DIGEST_CHANGED_WORD 6
#define SYMPA_FLAG_DIGEST_FORCE 1
int sympa_table_digest(const void* data, int64_t bytes, void* state, int flags, void* stream);
int sympa_table_pack_refresh(const double* table, int64_t num_rows, int n, int model, void* pack, int64_t pack_bytes,
                             void* digest_state, int flags, int32_t* status, void* stream);
int sympa_model_forward_packed(const void* pack, int64_t pack_bytes, int64_t num_rows, int n, const int64_t* src,
                               int64_t src_stride, const int64_t* dst, int64_t dst_stride, int64_t b, int model, int metric,
                               const double* metric_w, double eps, const double* scale, double scale_coef, double* out,
                               int32_t* status, int flags, void* stream);
int sympa_model_forward_batches_packed(const void* pack, int64_t pack_bytes, int64_t num_rows, int n,
                                       const int64_t* const* triplets, int64_t stride, const int64_t* b, int num_batches,
                                       int model, int metric, const double* metric_w, double eps, const double* scale,
                                       double scale_coef, double* const* out, int32_t* status, int flags, void* stream);

/* Measurement aid (bench.py `clock`): stamps the shader-cycle counter and the constant 100 MHz counter of SYMPA_CLOCK_STAMP_BLOCKS
 * one-wave blocks (several per CU) into out[block] = {s_memtime, s_memrealtime, XCC_ID | HW_ID << 8} (3 x uint64 per block,
 * device memory).  Two stamps around a stream-ordered region give the shader clock the chip HELD over that region:
 * d(s_memtime) / d(s_memrealtime) x 100 MHz between stamps taken on the SAME CU (bits 8..15 of HW_ID: CU, SH, SE; the counters of
 * different CUs are not comparable) -- what the fp64-issue roof of the timed kernels has to be priced at (the chip lowers
 * its clock under sustained load).  Not part of the reference's API; no product path calls it. */
#define SYMPA_CLOCK_STAMP_BLOCKS 2048
int sympa_clock_stamp(void* out, void* stream);

/* Block of rows of the all-pairs distance matrix that Runner.build_distance_matrix (sympa/runner.py:142-154)
 * assembles with N calls of Model.forward over N pairs each (for the mAP metric, sympa/metrics.py:39-63):
 *   out[(i - row_begin) * num_rows + j] = Model.forward((i, j)),  i in [row_begin, row_begin + row_count),  j in [0, N)
 * The diagonal is exactly 0 (the reference overwrites it with 0, runner.py:152).  For the FULL matrix (row_begin = 0,
 * row_count = num_rows) of dims >= 3 the symmetry d(i, j) = d(j, i) is used: the pairs i <= j are evaluated and every
 * value is stored at (i, j) and (j, i) (SYMPA_FLAG_NO_SYMMETRY evaluates both orders; a row block always does). */
int sympa_all_pairs_dist(const double* table, int64_t num_rows, int n, int64_t row_begin, int64_t row_count, int model,
                         int metric, const double* metric_w, double eps, const double* scale, double scale_coef,
                         double* out, int32_t* status, int flags, void* stream);

/* The same matrix with per-point factor reuse (dims <= SYMPA_MAX_DIMS_ALL_PAIRS_PACKED): every point is factored once
 * into `workspace` (sympa_all_pairs_workspace_bytes(num_rows, n, model) bytes of device memory owned by the caller,
 * overwritten; 0 = this build has no packed kernel for these dims), the pair kernel then needs neither a Cholesky nor
 * a gather, and for the FULL matrix (row_begin = 0, row_count = num_rows) of dims >= 3 only the pairs i <= j are
 * evaluated and each value is stored at (i, j) and (j, i) -- out is exactly symmetric -- unless SYMPA_FLAG_NO_SYMMETRY
 * is given; the diagonal is exactly 0 in every mode.  Same distances as
 * sympa_all_pairs_dist to rounding (1e-12 relative). */
int64_t sympa_all_pairs_workspace_bytes(int64_t num_rows, int n, int model);
int sympa_all_pairs_dist_packed(const double* table, int64_t num_rows, int n, int64_t row_begin, int64_t row_count,
                                int model, int metric, const double* metric_w, double eps, const double* scale,
                                double scale_coef, double* out, void* workspace, int64_t workspace_bytes,
                                int32_t* status, int flags, void* stream);

/* workspace (every backward entry below; caller-owned device scratch, 16-byte aligned, may be NULL): with at least
 * sympa_siegel_backward_workspace_bytes(b, n, model) bytes the SPLIT backward runs where it is built (dims 5..8, csrc/
 * siegel_bwd_split_kernel.hpp): ONE PAIR PER LANE in two kernels -- (1) factors, E, H = E^H E, Householder + QL with eigenvectors,
 * metric value, fused loss, Hbar = V diag(phi) V^H and K = Hbar H scaled by go * scale into the workspace ([entry][pair] layout);
 * (2) factors and E again from the table rows, Ebar = 2 E Hbar, G = Ebar E^H / 2, the solves and congruences, scatter / rows.  The
 * whole adjoint in one lane spills ~1100 registers at n = 8 and the eight-lanes-per-pair kernel repeats the scalar QL in the lanes
 * of a pair; the split form does neither (fused step, upper n = 8, 262 144 pairs: 1.42 -> 0.81 ms, 0.68 ms on batches sorted by source row).  It is the default where it
 * measured faster (upper model, dims 7, 8); SYMPA_FLAG_SPLIT runs it for every model and dims 5..8.  NULL / too small /
 * SYMPA_FLAG_GENERIC / SYMPA_FLAG_COOP: the other kernels, as before (same results to ~1e-11 relative: the split form takes the
 * spectral weights from the QL's eigenvalues instead of the Rayleigh quotients ||E v_i||^2).  Returns 0 where no kernel uses one.
 * Replaces the reference's autograd through siegel_manifold.py:41-72 (runner.py:105). */
int64_t sympa_siegel_backward_workspace_bytes(int64_t b, int n, int model);

/* Backward of manifold.dist for pre-gathered points: what torch autograd computes through
 * siegel_manifold.py:41-72 / bounded_domain.py:27-39 when runner.py:105 calls loss.backward().
 *   grad_out          [b] fp64 dLoss/d(dist)
 *   grad_z1, grad_z2  [b, 2, n, n] fp64, written (symmetric matrices, like the reference's gradients)
 *   grad_w            [n] fp64 gradient of the wsum weights, ACCUMULATED (zero it first), or NULL
 */
int sympa_siegel_dist_bwd(const double* z1, const double* z2, const double* grad_out, int64_t b, int n, int model,
                          int metric, const double* metric_w, double eps, double* grad_z1, double* grad_z2,
                          double* grad_w, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream);

/* Backward of Model.forward (sympa/model.py:16-41 under runner.py:105): recomputes the distances and
 * ACCUMULATES (fp64 atomics; zero the buffers first)
 *   grad_table  [num_rows, 2, n, n]  the dense embeds.grad the reference's DDP all-reduces (train.py:59)
 *   grad_w      [n]  wsum weights (or NULL),      grad_scale  [1]  model scale (or NULL)
 * `out` (may be NULL) receives the forward values [b] again.
 */
int sympa_model_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                         const int64_t* dst, int64_t dst_stride, int64_t b, int model, int metric,
                         const double* metric_w, double eps, const double* scale, double scale_coef,
                         const double* grad_out, double* grad_table, double* grad_w, double* grad_scale,
                         double* out, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream);

/* One fused training step of the reference's loop (sympa/runner.py:98-105 with sympa/losses.py:10-19):
 *   d = Model.forward(triplets);  loss = loss_scale * sum |(d / graph_dist)^2 - 1|;  loss.backward()
 * in ONE kernel: distances, loss (accumulated into loss[0]), and the gradients of the table, of the wsum
 * weights and of the scale (all ACCUMULATED: zero them first, or keep accumulating over grad-accum steps;
 * loss_scale = 1 / grad_accum_steps, runner.py:104).  graph_dist: [b] fp64.  `out` (may be NULL): d.
 */
int sympa_model_loss_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                              const int64_t* dst, int64_t dst_stride, const double* graph_dist, int64_t b, int model,
                              int metric, const double* metric_w, double eps, const double* scale, double scale_coef,
                              double loss_scale, double* loss, double* grad_table, double* grad_w, double* grad_scale,
                              double* out, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream);

/* The same fused step with the table gradient left in per-pair form (no scatter): grad_src_rows[i] / grad_dst_rows[i]
 * ([b, 2, n, n] each, written) are the gradient rows of table[src[i]] / table[dst[i]] contributed by pair i.  This is
 * the message of the touched-row gradient exchange of data-parallel training (sympa_amd/distributed.py): 2 b rows per
 * rank instead of the dense [num_rows, 2, n, n] tensor the reference's DDP all-reduces (train.py:59) -- smaller
 * whenever 2 x global batch < num_rows.  sympa_scatter_add_rows then accumulates any list of such rows,
 *     grad_table[idx[r * idx_stride]] += alpha * rows[r],   r in [0, count)      (alpha = 1 / world for DDP's mean),
 * into the dense gradient the optimiser step reads (fp64 atomics; rows of 2 n^2 doubles). */
int sympa_model_loss_backward_rows(const double* table, int64_t num_rows, int n, const int64_t* src,
                                   int64_t src_stride, const int64_t* dst, int64_t dst_stride, const double* graph_dist,
                                   int64_t b, int model, int metric, const double* metric_w, double eps,
                                   const double* scale, double scale_coef, double loss_scale, double* loss,
                                   double* grad_src_rows, double* grad_dst_rows, double* grad_w, double* grad_scale,
                                   double* out, int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream);
int sympa_scatter_add_rows(const double* rows, const int64_t* idx, int64_t idx_stride, int64_t count, int n,
                           int64_t num_rows, double alpha, double* grad_table, int32_t* status, void* stream);
/* the same for rows of any length (spd: row_doubles = n^2) */
int sympa_scatter_add_flat_rows(const double* rows, const int64_t* idx, int64_t idx_stride, int64_t count, int row_doubles,
                                int64_t num_rows, double alpha, double* grad_table, int32_t* status, void* stream);

/* ---- optimiser-side manifold operations over table rows (one [2,n,n] point per row) --------------------
 * egrad2rgrad: UpperHalfManifold.egrad2rgrad (sympa/manifolds/upper_half.py:25-40, Y G Y on both planes) /
 *              BoundedDomainManifold.egrad2rgrad (sympa/manifolds/bounded_domain.py:41-53, A G A, A = I - conj(Z) Z).
 */
int sympa_egrad2rgrad(const double* z, const double* u, int64_t b, int n, int model, double* out, void* stream);

/* inner(z, u, u) per row: UpperHalfManifold.inner (sympa/manifolds/upper_half.py:68-91, tr[y^-1 u y^-1 conj(u)]) /
 * BoundedDomainManifold.inner (sympa/manifolds/bounded_domain.py:86-116) for any complex tangent vector u:
 * out[i] = squared Riemannian norm of u[i] at z[i].  The second moment of geoopt's RiemannianAdam (train.py:69-70). */
int sympa_tangent_sqnorm(const double* z, const double* u, int64_t b, int n, int model, double* out, int32_t* status,
                         void* stream);

/* projx: SiegelManifold.projx + UpperHalfManifold.projx (siegel_manifold.py:130-137, upper_half.py:42-66,
 * csym_math.py:252-278): symmetrise, clamp the eigenvalues of Im z at eps, rows already inside are left
 * untouched; BoundedDomainManifold.projx as intended by bounded_domain.py:55-84: clamp the Takagi values
 * at 1 - eps.  projected_count[0] += number of rows that were moved (manifold.projected_points).
 * outside_word: caller-owned device int32 scratch word, one per call in flight (dims >= 7: the lanes-per-row kernel counts
 * the rows that left the eps-interior in it and the exact eigenvalue clamp runs gated on that count; the library keeps no
 * state of its own -- re-entrant across streams, devices and threads).  NULL: every row goes through the exact clamp
 * (one row per lane; same result, slower at dims >= 7).  Ignored at dims <= 6. */
int sympa_projx(const double* z, int64_t b, int n, int model, double eps, double* out, int32_t* projected_count,
                int32_t* status, int32_t* outside_word, void* stream);

/* One RiemannianSGD step over the whole table, in place (geoopt.optim.RiemannianSGD.step with momentum 0,
 * the optimiser train.py:66-68 builds; geoopt is absent from the reference tree, semantics restated):
 *     table <- retr(table, -lr * egrad2rgrad(table, grad + weight_decay * table)),  retr(x,u) = projx(x + u)
 * (sympa/manifolds/siegel_manifold.py:74-87).  outside_word: as in sympa_projx. */
int sympa_rsgd_step(double* table, const double* grad, int64_t num_rows, int n, int model, double lr,
                    double weight_decay, double eps, int32_t* projected_count, int32_t* status, int32_t* outside_word,
                    void* stream);

/* One RiemannianAdam step over the whole table, in place, ONE launch (geoopt.optim.RiemannianAdam as train.py:69-70 builds it:
 * `--optim radam`, eps = 1e-7, stabilize = None; geoopt is absent from the reference tree, the step is restated from
 * geoopt/optim/radam.py):  g = egrad2rgrad(x, grad + weight_decay x);  exp_avg <- beta1 exp_avg + (1 - beta1) g;
 * exp_avg_sq[row] <- beta2 exp_avg_sq[row] + (1 - beta2) inner(x, g, g);
 * x <- projx(x - lr (exp_avg / (1 - beta1^t)) / (sqrt(exp_avg_sq / (1 - beta2^t)) + eps_adam)).
 * exp_avg: [num_rows, 2, n, n], exp_avg_sq: [num_rows]; bias_pows: DEVICE words {beta1^t, beta2^t} of this step (the caller
 * advances them on the device, so a captured launch never carries a step count).  dims 1..6 (one row per lane); larger
 * tables use sympa_egrad2rgrad / sympa_tangent_sqnorm / sympa_projx. */
int sympa_radam_step(double* table, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t num_rows, int n, int model,
                     double lr, double beta1, double beta2, double eps_adam, double weight_decay, const double* bias_pows,
                     double eps, int32_t* projected_count, int32_t* status, void* stream);

/* The gradient clip of the reference's loop (torch.nn.utils.clip_grad_norm_, sympa/runner.py:115) folded into the step:
 * sympa_sqnorm_accum adds sum(x^2) to acc[0] (device; call it once per gradient tensor after zeroing acc), and
 * sympa_rsgd_step_clipped is sympa_rsgd_step with every gradient row scaled by
 *     min(1, max_norm / (sqrt(total_sqnorm[0]) + 1e-6))
 * as it is loaded (the gradient buffer itself is left unscaled). */
int sympa_sqnorm_accum(const double* x, int64_t count, double* acc, void* stream);
/* The plain SGD step of the parameters that live on no manifold (the model's scale: RiemannianSGD on a Euclidean
 * parameter, geoopt/optim/rsgd.py) with the same folded clip: p <- p - lr * (coef * grad + weight_decay * p). */
int sympa_sgd_step_clipped(double* p, const double* grad, int64_t count, double lr, double weight_decay,
                           const double* total_sqnorm, double max_norm, void* stream);
int sympa_rsgd_step_clipped(double* table, const double* grad, int64_t num_rows, int n, int model, double lr,
                            double weight_decay, double eps, const double* total_sqnorm, double max_norm,
                            int32_t* projected_count, int32_t* status, int32_t* outside_word, void* stream);

/* The backward half of a training step inside a replayed hipGraph (sympa/runner.py:98-105), dims 1..16 (wave_partials: dims
 * 1..8).  Like
 * sympa_model_loss_backward (grad_table given: fp64-atomic scatter into the dense gradient) or
 * sympa_model_loss_backward_rows (grad_rows [2 b, 2, n, n] given: rows [0, b) belong to src, [b, 2 b) to dst), plus
 *   step_counter   device int64 c (may be NULL = 0): the launch processes pairs [c b, (c + 1) b) of src / dst / graph_dist,
 *                  i.e. an epoch's shuffled triplets (train.py:105-110) stay in ONE buffer, sympa_rsgd_step_fused increments
 *                  c, and the replayed graph needs no per-step copy of the batch;
 *   wave_partials  (rows form; may be NULL) [ceil(b / 64)][2 + n] fp64, WRITTEN: per-wavefront sums of the loss, of
 *                  d loss / d scale and of d loss / d w_k, which sympa_segment_sum_rows adds to loss / grad_scale / grad_w in
 *                  a fixed order -- together with the row sums below this makes a training step bitwise reproducible
 *                  (the reference's CPU autograd is; fp64 atomics are not).  With wave_partials loss / grad_* are untouched. */
int sympa_model_train_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                               const int64_t* dst, int64_t dst_stride, const double* graph_dist, int64_t b,
                               const int64_t* step_counter, int model, int metric, const double* metric_w, double eps,
                               const double* scale, double scale_coef, double loss_scale, double* loss, double* grad_table,
                               double* grad_rows, double* grad_w, double* grad_scale, double* wave_partials, int32_t* status,
                               void* workspace, int64_t workspace_bytes, int flags, void* stream);

/* Deterministic counterpart of sympa_scatter_add_flat_rows ("sort-by-row + segmented sum"):
 *   grad_table[r] (= | +=, by `accumulate`) alpha * sum_{p in [rowptr[r], rowptr[r + 1])} rows[order[p]]     r = 0 .. num_rows - 1
 * added in the order of `order`, no atomics: every element is written by exactly one thread, rows nobody touched are
 * written as 0 when accumulate = 0 (no zeroing pass).  order: int32 slot numbers (rows of `rows`, row_doubles each)
 * sorted by table row -- stable, so the order inside a row is the batch order; rowptr: int32 [num_rows + 1].  With
 * step_counter c (device int64, may be NULL) the lists of batch c are used: order + c * order_stride, rowptr + c *
 * (num_rows + 1) (sympa_amd/train_step.py builds them for a whole epoch with one sort).  wave_partials (may be NULL): see
 * sympa_model_train_backward; partial_stride = 2 + n (doubles per wavefront), num_weights = n for the wsum metric, else 0
 * (then grad_w may be NULL); grad_scale may be NULL.  sq_partials (may be NULL; needs wave_partials): [sympa_segment_sum_
 * partials(num_rows, row_doubles)] fp64, WRITTEN: one sum of squares of the finished gradient per block in a fixed tree, the
 * last one for the scale / weight gradients -- sympa_rsgd_step_fused adds them in index order instead of making its own pass. */
int64_t sympa_segment_sum_partials(int64_t num_rows, int row_doubles);
int sympa_segment_sum_rows(const double* rows, const int32_t* order, const int32_t* rowptr, int64_t num_rows, int row_doubles,
                           int64_t order_stride, const int64_t* step_counter, double alpha, int accumulate, double* grad_table,
                           const double* wave_partials, int64_t num_waves, int partial_stride, int num_weights, double* loss,
                           double* grad_scale, double* grad_w, double* sq_partials, void* stream);

/* The optimiser side of one training step as ONE launch (sympa/runner.py:113-118: clip_grad_norm_(parameters, max_norm),
 * optimizer.step(), zero_grad()), dims 1..6, tables of at most (CUs x 256) rows (a grid barrier inside: every block must be
 * resident; otherwise SYMPA_ERR_UNSUPPORTED_DIMS -- use sympa_sqnorm_accum + sympa_rsgd_step_clipped + sympa_sgd_step_clipped):
 *   total = ||grad||^2 + sum_k ||extra_grad[k]||^2, summed in a fixed order (bitwise reproducible);
 *   coef = max_norm > 0 ? min(1, max_norm / (sqrt(total) + 1e-6)) : 1;
 *   table <- retr(table, -lr * egrad2rgrad(table, coef * grad + weight_decay * table))           (sympa_rsgd_step)
 *   extra_param[k] <- extra_param[k] - extra_lr[k] * (coef * extra_grad[k] + extra_weight_decay[k] * extra_param[k])
 *   zero_grads != 0: grad and extra_grad[k] are zeroed (the next step's backward accumulates into them);
 *   step_counter (device int64, may be NULL) += 1: the batch index the next sympa_model_loss_backward_indexed call reads.
 * extra_*: up to 2 parameters without a manifold (the model's scale, the wsum weights), 1..64 elements each.
 * sq_partials (may be NULL): partial sums of squares that sympa_segment_sum_rows left -- then `total` is their sum in index
 * order and the kernel makes no pass over the gradient and needs no barrier (the deterministic training step).
 * workspace: sympa_rsgd_step_fused_workspace_bytes(num_rows) bytes of device memory, ZERO before the first call (the
 * kernel leaves it ready for the next one); one workspace per concurrently running step. */
int64_t sympa_rsgd_step_fused_workspace_bytes(int64_t num_rows);
int sympa_rsgd_step_fused(double* table, double* grad, int64_t num_rows, int n, int model, double lr, double weight_decay,
                          double eps, double max_norm, int zero_grads, double* const* extra_param, double* const* extra_grad,
                          const int* extra_count, const double* extra_lr, const double* extra_weight_decay, int num_extra,
                          void* workspace, int64_t workspace_bytes, const double* sq_partials, int num_sq_partials,
                          int64_t* step_counter, int32_t* projected_count, int32_t* status, void* stream);

/* The same launch for `--optim radam` (train.py:69-70: geoopt.optim.RiemannianAdam, eps = 1e-7): clip, one RiemannianAdam
 * step of the table (sympa_radam_step's formula), the ordinary Adam step of the plain parameters, zero_grad, step counter.
 * bias_pows / extra_bias_pows[k]: device words {beta1^t, beta2^t} as the PREVIOUS step left them (1, 1 before the first):
 * the kernel uses beta * word as this step's power and stores it back, so a replayed launch carries no step count.
 * Same limits and workspace as sympa_rsgd_step_fused. */
int sympa_radam_step_fused(double* table, double* grad, double* exp_avg, double* exp_avg_sq, double* bias_pows, int64_t num_rows,
                           int n, int model, double lr, double beta1, double beta2, double eps_adam, double weight_decay,
                           double eps, double max_norm, int zero_grads, double* const* extra_param, double* const* extra_grad,
                           double* const* extra_exp_avg, double* const* extra_exp_avg_sq, double* const* extra_bias_pows,
                           const int* extra_count, const double* extra_lr, const double* extra_weight_decay, int num_extra,
                           void* workspace, int64_t workspace_bytes, const double* sq_partials, int num_sq_partials,
                           int64_t* step_counter, int32_t* projected_count, int32_t* status, void* stream);

/* ---- SPD model (manifold "spd": geoopt.manifolds.SymmetricPositiveDefinite, sympa/embeddings.py:6,70-72,142) ----
 * Points are [n, n] fp64 symmetric positive definite matrices (upper triangle read), n <= 16.
 * dist = || log(x^-1/2 y x^-1/2) ||_F  (geoopt's default affine-invariant metric; geoopt is absent from the
 * reference tree: parity unpinned).  sympa_spd_model_forward is Model.forward for that model
 * (sympa/model.py:16-41) with a [num_rows, n, n] table.  flags: 0 or SYMPA_FLAG_GENERIC.
 * 6 <= n <= 16 run the sixteen-lanes-per-pair kernel (csrc/spd_coop.hpp: one instantiation per n, lanes r >= n of a group
 * are phantoms; factorisation, solves and the first n - 10 Householder steps sixteen lanes per pair, the trailing block of at
 * most 10 x 10 and the QL iteration one pair per lane), n <= 5 the runtime-n one-lane-per-pair kernel. */
int sympa_spd_dist_fwd(const double* x, const double* y, int64_t b, int n, double* out, int32_t* status, int flags,
                       void* stream);
int sympa_spd_model_forward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                            const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale, double scale_coef,
                            double* out, int32_t* status, int flags, void* stream);

/* Packed table of the spd model, n = 6..16 (csrc/spd.hip; the Siegel counterpart is sympa_table_pack): one row of n (n + 1)
 * doubles per point -- an n x n image whose upper triangle (diagonal included) is the point and whose strict lower triangle is the
 * UNIT factor Lh of x = Lh D Lh^T, then the n values D^-1/2 -- made once per table version.  sympa_spd_model_forward_packed is
 * sympa_spd_model_forward reading it: the sixteen-lanes-per-pair kernel without its factorisation (~310 of the ~1 170 VALU
 * instructions of a round of four pairs), the first point's image read twice (symmetric rows for y - x, then its factor rows), the
 * second point's upper triangle only.  Same values to rounding (the factor is the one the dense kernel computes).  A point that
 * is not positive definite is reported by the pack call (SYMPA_ST_NOT_PD) and every pair it enters later comes out NaN and
 * flagged.  sympa_spd_table_pack_bytes returns 0 where no packed kernel exists (n < 6).  Replaces the per-pair factorisation inside
 * geoopt's SymmetricPositiveDefinite.dist (sympa/embeddings.py:70-72,142; parity UNPINNED like every spd entry). */
int64_t sympa_spd_table_pack_bytes(int64_t num_rows, int n);
int sympa_spd_table_pack_refresh(const double* table, int64_t num_rows, int n, void* pack, int64_t pack_bytes, void* digest_state,
                                 int flags, int32_t* status, void* stream);
int sympa_spd_table_pack(const double* table, int64_t num_rows, int n, void* pack, int64_t pack_bytes, int32_t* status,
                         void* stream);
int sympa_spd_model_forward_packed(const void* pack, int64_t pack_bytes, int64_t num_rows, int n, const int64_t* src,
                                   int64_t src_stride, const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale,
                                   double scale_coef, double* out, int32_t* status, int flags, void* stream);

/* ---- SPD model, training path (PARITY UNPINNED like the forward; formulas restated from geoopt's published source,
 * pinned by 50-digit finite differences, tests/golden/spd_n*.npz) ----
 * sympa_spd_backward_rows: backward of SymmetricPositiveDefinite.dist / of Model.forward for the spd model
 * (sympa/model.py:16-41 under runner.py:105).  Pair i = (x[src[i]], y[dst[i]]) (src = dst = NULL: rows i of x and y).
 * Either grad_out [b] (dLoss/d out) or graph_dist [b] is given; with graph_dist the AverageDistortionLoss of
 * sympa/losses.py:10-19 is fused in: loss[0] += loss_scale * sum |(out / graph_dist)^2 - 1|.
 * grad_x_rows / grad_y_rows [b, n, n] are WRITTEN with the (symmetric) gradient rows of the two points of each pair;
 * accumulate them into a table gradient with sympa_scatter_add_flat_rows.  grad_scale [1] accumulated, out [b] optional.
 * n >= 3 runs sixteen lanes per pair (csrc/spd_coop_bwd.hpp); flags: 0, SYMPA_FLAG_GENERIC (one lane per pair over scratch)
 * or SYMPA_FLAG_COOP (see above).  The row operations below run sixteen lanes per row for n >= 3 (csrc/spd_coop_table.hpp;
 * environment SYMPA_SPD_TABLE_GENERIC=1 keeps the one-row-per-lane kernel for measurements; projx always uses it). */
int sympa_spd_backward_rows(const double* x, const double* y, int64_t num_rows, int n, const int64_t* src,
                            int64_t src_stride, const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale,
                            double scale_coef, const double* grad_out, const double* graph_dist, double loss_scale,
                            double* loss, double* grad_x_rows, double* grad_y_rows, double* grad_scale, double* out,
                            int32_t* status, void* workspace, int64_t workspace_bytes, int flags, void* stream);
/* workspace (both backward entries; caller-owned device scratch, 16-byte aligned, may be NULL): with at least
 * sympa_spd_backward_workspace_bytes(b, n) bytes the THREE-KERNEL backward runs where it is built (n = 9..16, csrc/
 * spd_coop_bwd3_kernel.hpp): Householder form sixteen lanes per pair, then eigenvalues (lockstep QL) and eigenvectors (inverse
 * iteration) ONE PAIR PER LANE with the vectors parked in the workspace, then the gradient rows sixteen lanes per pair again --
 * 3x the kernel that runs the QL with accumulated rotations in the sixteen-lanes layout; pairs with a block of more than four
 * close eigenvalues (y = c x) are finished by that kernel in a second launch behind it.  NULL, a size of 0 or SYMPA_FLAG_COOP /
 * SYMPA_FLAG_GENERIC: the other kernels, as before.  sympa_spd_backward_workspace_bytes returns 0 where no kernel uses one. */
int64_t sympa_spd_backward_workspace_bytes(int64_t b, int n);
/* The same with the scatter inside the kernel (n >= 3, index lists required): the gradient rows of each pair are added
 * into grad_table [num_rows, n, n] with fp64 atomics, consecutive lanes on consecutive doubles -- one launch for
 * Model.forward + AverageDistortionLoss + backward + embedding-gradient accumulation (runner.py:98-118 for the spd model). */
int sympa_spd_loss_backward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                            const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale, double scale_coef,
                            const double* grad_out, const double* graph_dist, double loss_scale, double* loss,
                            double* grad_table, double* grad_scale, double* out, int32_t* status, void* workspace, int64_t workspace_bytes, int flags,
                            void* stream);
/* geoopt SymmetricPositiveDefinite.egrad2rgrad (x sym(u) x), .projx (V |lambda| V^T of sym(x); projected_count += rows with
 * a negative eigenvalue), and one geoopt.optim.RiemannianSGD step over the table in place,
 *     x <- retr(x, -lr * egrad2rgrad(x, grad + weight_decay * x)),   retr(x, u) = sym(x + u + u x^-1 u / 2),
 * with the gradient clip of runner.py:115 folded in when total_sqnorm (device, squared total norm) is given. */
int sympa_spd_egrad2rgrad(const double* x, const double* u, int64_t b, int n, double* out, void* stream);
int sympa_spd_projx(const double* x, int64_t b, int n, double* out, int32_t* projected_count, int32_t* status, void* stream);
int sympa_spd_rsgd_step(double* table, const double* grad, int64_t num_rows, int n, double lr, double weight_decay,
                        const double* total_sqnorm, double max_norm, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SYMPA_HIP_H */
