// gfx950 kernel + C-ABI of the SPD model (spd_math.hpp): runtime-n, per-lane scratch, 64-thread blocks.
#include "siegel_common.hpp"
#include "spd_coop.hpp"
#include "spd_math.hpp"
#include "table_digest.hpp"

namespace {
using namespace sympa_hip;

// `row`: doubles per table row -- n * n, or n (n + 1) over a PACKED table (sympa_spd_table_pack: the upper triangle of its n x n image
// is the point, which is all spd_pair_distance reads): the one-lane kernel a demoted packed instantiation falls back to
__global__ __launch_bounds__(64) void spd_dist_kernel(const DistArgs a, const int n, const int64_t row) {
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

// n = 16: sixteen lanes per pair for the factorisation, the solves and the first six Householder steps, one lane per pair
// for the trailing 10 x 10 block and the QL iteration (spd_coop.hpp).  One wave per block, 64 pairs per wave, 19.2 KB of LDS
// (the 16 KB tile + 3.2 KB for the hand-over), two waves per SIMD.  Lane 16 g + t owns pair 4 t + g of the wave's 64.
// M < 16 (6 <= n < 16, one instantiation per n): the row-per-lane routines are templates over the matrix size, lanes
// r >= M of a group are phantoms (spd_coop.hpp), the time goes with M^2.  Rows are n*n*8 bytes then, not the 2 KB image
// the DMA tile is made for, so each lane loads the elements of its row itself (upper triangle: (min, max)).
// two waves per SIMD (256 registers) up to this size of the handed-over block, one beyond
// PACKED (round 5; C-ABI sympa_spd_table_pack / sympa_spd_model_forward_packed): base1 = base2 = the packed table, one row of
// n (n + 1) doubles per point -- an n x n image whose upper triangle (with the diagonal) is the point and whose strict lower
// triangle is its UNIT factor Lh (X = Lh D Lh^T), followed by the n values D^-1/2.  The factorisation (ldl_rows: sixteen steps of
// pivot capture + DPP updates, ~310 of the ~1 170 VALU instructions of a round) then happens once per table version in the
// pack kernel instead of once per pair; a round reads the first point's image twice (its symmetric rows for Y - X, then its
// factor rows in place of them) and the second point's upper triangle only.
template <int M, bool PACKED = false>
__global__ __launch_bounds__(64, (spd_coop::trailing_block<M>() <= 10 ? 2 : 1)) void spd16_coop_kernel(const DistArgs a) {
    using namespace spd_coop;
    constexpr bool PADDED = M < N;      // historical name: "not the 2 KB image of n = 16"
    constexpr int n = M;
    constexpr int TB = trailing_block<M>();
    constexpr unsigned ROWB = PACKED ? (unsigned)(M * (M + 1) * 8) : 2048u;     // bytes per table row of the image path (M = 16)
    constexpr unsigned ROWD = PACKED ? (unsigned)(M * (M + 1)) : (unsigned)(M * M);   // doubles per table row
    __shared__ __attribute__((aligned(16))) char tile[LDS_BYTES];
    __shared__ __attribute__((aligned(16))) double rdl_lds[PACKED && !PADDED ? GROUPS_PER_WAVE * N : 2];
    __shared__ __attribute__((aligned(16))) double hand_all[GROUPS_PER_WAVE * (TB >= 3 ? TB * TB : 2) + (TB >= 3 ? TB + (TB & 1) : 0)];    // + one dummy row
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15;
    double* const hand = hand_all + g * (TB >= 3 ? TB * TB : 2);
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

    // DMA: instruction q of a round fills LDS slots [64 q, 64 q + 64) = half h = q & 1 of matrix image q >> 1;
    // slot s of an image holds columns {2c, 2c+1} of row s / 8 with c = (s & 7) ^ swizzle(row)  (tile_slot)
    unsigned voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rr = h * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((rr >> 1) & 7);
        voff[h] = (unsigned)(rr * 128 + c * 16);
    }
    // byte offset of element (r, j) of my pair's X image (upper triangle only: (min, max))
    // (two 14-bit offsets per register: the sixteen of them would otherwise hold sixteen registers through the kernel)
    unsigned eoff2[(M + 1) / 2];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int lo = r < j ? r : j, hi = r < j ? j : r;
        const unsigned e = (unsigned)(g * 4096 + tile_slot(lo, hi >> 1) * 16 + (hi & 1) * 8);
        if (j & 1) eoff2[j / 2] |= e << 16;
        else eoff2[j / 2] = e;
    }
    auto issue = [&](const int t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gg = q >> 2, side = (q >> 1) & 1, h = q & 1;
            const int row = __builtin_amdgcn_readlane(side ? row2 : row1, 16 * gg + t);
            const char* src = reinterpret_cast<const char*>(side ? a.base2 : a.base1) + (size_t)(unsigned)row * ROWB + voff[h];
            // (fetching only the chunks of the SECOND point that reach its upper triangle -- 84 of 128 -- was measured: no gain,
            // packed / dense 0.904 against 0.899 with whole images, profiles/r05_spd_packed_forward.txt)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(tile + q * 1024), 16, 0, 0);
        }
        if constexpr (PACKED) {          // D^-1/2 of the four first points: 128 bytes behind each image
#pragma unroll
            for (int gg = 0; gg < GROUPS_PER_WAVE; ++gg) {
                const int row = __builtin_amdgcn_readlane(row1, 16 * gg + t);
                const char* src = reinterpret_cast<const char*>(a.base1) + (size_t)(unsigned)row * ROWB + 2048u + (unsigned)(lane & 7) * 16u;
                if (lane < 8) __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(rdl_lds + gg * N), 16, 0, 0);
            }
        }
    };
    double d[M], e2[M], blk[packed_len<TB>()];
#pragma unroll
    for (int k = 0; k < M; ++k) { d[k] = 0.0; e2[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < packed_len<TB>(); ++k) blk[k] = 0.0;
    bool ok = true;
    if constexpr (!PADDED) issue(0);
    for (int t = 0; t < ROUNDS; ++t) {
        double x[M], y[M];
        if constexpr (PADDED) {
            const int rowx = __builtin_amdgcn_ds_bpermute(4 * (16 * g + t), row1);     // rows of my group's pair
            const int rowy = __builtin_amdgcn_ds_bpermute(4 * (16 * g + t), row2);
            const double* px = a.base1 + (size_t)(unsigned)rowx * ROWD;
            const double* py = a.base2 + (size_t)(unsigned)rowy * ROWD;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const int lo = r < j ? r : j, hi = r < j ? j : r;
                const int e = (hi < n) ? lo * n + hi : 0;          // a phantom lane (r >= M) reads element 0
                x[j] = px[e];
                y[j] = py[e];
            }
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0): this round's images have landed
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < M; ++j) {
                unsigned pk = eoff2[j / 2];
                asm volatile("" : "+v"(pk));        // unpack here, every round: hoisted out of the loop it is sixteen registers again
                const unsigned e = (j & 1) ? (pk >> 16) : (pk & 0xffffu);
                x[j] = *reinterpret_cast<const double*>(tile + e);
                y[j] = *reinterpret_cast<const double*>(tile + 2048 + e);
            }
        }
        if constexpr (TB >= 3) {
            // the trailing block of the previous round's pair, behind this round's loads
            wave_lds_fence();
            if (t > 0 && r == t - 1) take_block<TB>(blk, hand);
        }
        double rdl, m[M];
        bool pd;
        if constexpr (PACKED) {
            // A = Y - X with the symmetric rows, then the first point's FACTOR rows (strict lower triangle of the same image:
            // element (r, j) itself) and D^-1/2 take the place of x -- what ldl_rows would have left there
#pragma unroll
            for (int j = 0; j < M; ++j) y[j] -= x[j];
            if constexpr (PADDED) {
                const int rowx = __builtin_amdgcn_ds_bpermute(4 * (16 * g + t), row1);
                const double* px = a.base1 + (size_t)(unsigned)rowx * ROWD;
#pragma unroll
                for (int j = 0; j < M; ++j) x[j] = px[(r < n) ? r * n + j : 0];
                rdl = (r < n) ? px[n * n + r] : 1.0;
            } else {
                // (derived from the lane id here, every round: hoisted out of the loop they are two more registers of a kernel that
                // sits at exactly 256)
                unsigned ln = (unsigned)lane;
                asm volatile("" : "+v"(ln));
                const unsigned base = (ln >> 4) * 4096u + (ln & 15u) * 128u;
                const unsigned sw = ((ln >> 1) & 7u) << 4;
#pragma unroll
                for (int c = 0; c < M / 2; ++c) {
                    const v2d q = *reinterpret_cast<const v2d*>(tile + base + (((unsigned)c << 4) ^ sw));
                    x[2 * c] = q.x;
                    x[2 * c + 1] = q.y;
                }
                rdl = rdl_lds[g * N + r];
            }
            pd = reduce_pair_front_factored(x, y, rdl, m, reinterpret_cast<double*>(tile) + g * TBUF, r);
        } else {
            pd = reduce_pair_front(x, y, rdl, m, reinterpret_cast<double*>(tile) + g * TBUF, r);
        }
        // the tile is free again (images and transpose consumed): fetch the next round behind the arithmetic
        __builtin_amdgcn_s_waitcnt(0xC07F);
        wave_lds_fence();
        if constexpr (!PADDED) if (t + 1 < ROUNDS) issue(t + 1);
        const bool keep = (r == t);
        ok = keep ? pd : ok;
        reduce_pair_back<M, TB>(m, x, rdl, r, keep, 0x0001000100010001ull << t, d, e2, hand,
                                hand_all + GROUPS_PER_WAVE * (TB >= 3 ? TB * TB : 2));
    }
    if constexpr (TB >= 3) {
        wave_lds_fence();
        if (r == ROUNDS - 1) take_block<TB>(blk, hand);
    }
    // one pair per lane from here: the rest of the tridiagonalisation (the trailing TB x TB block of my pair), ...
    if constexpr (TB >= 3) sympa::tridiag_packed<TB>(blk, d + (M - TB), e2 + (M - TB));
    // ... QL on the tridiagonal forms, then the norm of the logarithms
    {   // The lockstep iteration deflates position 0 first; started from the END of the Householder form (the block that was
        // reduced last) the wave needs 6 % fewer sweeps (simulated on the bench table: 372 -> 350 element-sweeps per wave)
        // and the kernel measures 1.2 % faster (profiles/r03_spd_forward_ab.txt).  A register renaming, no instructions.
        double dr[M], er[M];
#pragma unroll
        for (int k = 0; k < M; ++k) { dr[k] = d[M - 1 - k]; er[k] = (k < M - 1) ? e2[M - 2 - k] : 0.0; }
#pragma unroll
        for (int k = 0; k < M; ++k) { d[k] = dr[k]; e2[k] = er[k]; }
    }
    const bool conv = sympa::tridiag_ql_lockstep<M, 0, true>(d, e2);      // (forward: eigenvalues only)
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        ok = ok && (d[k] > -1.0);
        const double lg = sympa::d_log1p_signed(d[k]);
        acc = sympa::d_fma(lg, lg, acc);
    }
    double out = sympa::d_sqrt(acc);
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

// the pack kernel: sixteen lanes per point, lane r owns row r; X = Lh D Lh^T by ldl_rows, row r of the packed image = the point's
// row right of (and on) the diagonal, the factor's row left of it, D^-1/2 behind the image (NaN when the point is not positive
// definite: every pair it enters then comes out NaN and is flagged)
template <int M>
__global__ __launch_bounds__(64) void spd_pack_kernel(const double* __restrict__ table, const int64_t num_rows, double* __restrict__ pack,
                                                      int32_t* status, const unsigned* guard) {
    using namespace spd_coop;
    if (guard != nullptr && __builtin_nontemporal_load(guard) == 0u) return;      // sympa_spd_table_pack_refresh: table unchanged
    const int lane = threadIdx.x;
    const int g = lane >> 4, r = lane & 15;
    const int64_t i = (int64_t)blockIdx.x * GROUPS_PER_WAVE + g;
    const int64_t ii = i < num_rows ? i : num_rows - 1;
    const double* px = table + ii * (M * M);
    double x[M], orig[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int lo = r < j ? r : j, hi = r < j ? j : r;
        x[j] = px[(hi < M) ? lo * M + hi : 0];
        orig[j] = x[j];
    }
    double rdl;
    const bool pd = ldl_rows(x, rdl);
    if (i < num_rows && r < M) {
        double* out = pack + i * (M * (M + 1));
#pragma unroll
        for (int j = 0; j < M; ++j) out[r * M + j] = (j < r) ? x[j] : orig[j];
        out[M * M + r] = pd ? rdl : __builtin_nan("");
    }
    if (status != nullptr) {
        const int bad = (i < num_rows && !pd && r == 0) ? 1 : 0;
        const unsigned long long m = __ballot(bad);
        if (m != 0ull && lane == 0) {
            atomicOr(&status[0], sympa::ST_NOT_PD);
            atomicAdd(&status[1], (int)__popcll(m));
        }
    }
}

int launch_spd(const DistArgs& a, int n, void* stream, const bool packed = false) {
    if (a.b < 0) return fail(SYMPA_ERR_BAD_ARG, "negative batch size");
    if (a.b == 0) return 0;
    if (a.base1 == nullptr || a.base2 == nullptr || a.out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (n < 1 || n > sympa::SPD_MAX_N) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd: dims outside [1, 16]");
    if (a.num_rows > (int64_t)0x7fffffff) return fail(SYMPA_ERR_BAD_ARG, "more than 2^31-1 table rows");
    const dim3 grid((unsigned)((a.b + 63) / 64));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n >= 6 && !(a.flags & SYMPA_FLAG_GENERIC) && !instance_fallback(SYMPA_FAMILY_SPD_FWD, 0, n)) {     // measured crossover against the one-lane-per-pair kernel: n = 6
        switch (n) {
#define SYMPA_SPD_COOP_CASE(MM) case MM: if (packed) hipLaunchKernelGGL((spd16_coop_kernel<MM, true>), grid, dim3(64), 0, s, a); \
                                         else hipLaunchKernelGGL((spd16_coop_kernel<MM, false>), grid, dim3(64), 0, s, a); break;
            SYMPA_SPD_COOP_CASE(6) SYMPA_SPD_COOP_CASE(7) SYMPA_SPD_COOP_CASE(8) SYMPA_SPD_COOP_CASE(9)
            SYMPA_SPD_COOP_CASE(10) SYMPA_SPD_COOP_CASE(11) SYMPA_SPD_COOP_CASE(12) SYMPA_SPD_COOP_CASE(13)
            SYMPA_SPD_COOP_CASE(14) SYMPA_SPD_COOP_CASE(15) SYMPA_SPD_COOP_CASE(16)
#undef SYMPA_SPD_COOP_CASE
            default: break;
        }
    } else {
        // (packed with the instantiation demoted by the self-check, sympa_set_instance_fallback: the one-lane kernel over the pack's
        // images -- round-5 advice: the packed entry used to fail here while the dense entry fell back)
        if (packed && n < 6) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd packed forward: dims 6..16");
        hipLaunchKernelGGL(spd_dist_kernel, grid, dim3(64), 0, s, a, n, packed ? (int64_t)n * (n + 1) : (int64_t)n * n);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

int sympa_spd_dist_fwd(const double* x, const double* y, int64_t b, int n, double* out, int32_t* status, int flags,
                       void* stream) {
    DistArgs a;
    std::memset(&a, 0, sizeof(a));
    a.base1 = x;
    a.base2 = y;
    a.b = b;
    a.num_rows = b;
    a.inv_scale_coef = 1.0;
    a.out = out;
    a.status = status;
    a.flags = flags;
    return launch_spd(a, n, stream);
}

int sympa_spd_model_forward(const double* table, int64_t num_rows, int n, const int64_t* src, int64_t src_stride,
                            const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale, double scale_coef,
                            double* out, int32_t* status, int flags, void* stream) {
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
    a.flags = flags;
    return launch_spd(a, n, stream);
}

int64_t sympa_spd_table_pack_bytes(int64_t num_rows, int n) {
    if (num_rows <= 0 || n < 6 || n > sympa::SPD_MAX_N) return 0;
    return num_rows * (int64_t)(n * (n + 1)) * 8;
}

namespace {
int spd_pack_any(const double* table, int64_t num_rows, int n, void* pack, int64_t pack_bytes, int32_t* status, const unsigned* guard,
                 void* stream) {
    if (n < 6 || n > sympa::SPD_MAX_N) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd packed table: dims 6..16");
    if (table == nullptr || num_rows <= 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (pack == nullptr || pack_bytes < sympa_spd_table_pack_bytes(num_rows, n) || (reinterpret_cast<uintptr_t>(pack) & 15))
        return fail(SYMPA_ERR_BAD_ARG, "packed table: a 16-byte aligned buffer of sympa_spd_table_pack_bytes(num_rows, n) bytes");
    const dim3 grid((unsigned)((num_rows + spd_coop::GROUPS_PER_WAVE - 1) / spd_coop::GROUPS_PER_WAVE));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* p = reinterpret_cast<double*>(pack);
    switch (n) {
#define SYMPA_SPD_PACK_CASE(MM) case MM: hipLaunchKernelGGL(spd_pack_kernel<MM>, grid, dim3(64), 0, s, table, num_rows, p, status, guard); break;
        SYMPA_SPD_PACK_CASE(6) SYMPA_SPD_PACK_CASE(7) SYMPA_SPD_PACK_CASE(8) SYMPA_SPD_PACK_CASE(9) SYMPA_SPD_PACK_CASE(10)
        SYMPA_SPD_PACK_CASE(11) SYMPA_SPD_PACK_CASE(12) SYMPA_SPD_PACK_CASE(13) SYMPA_SPD_PACK_CASE(14) SYMPA_SPD_PACK_CASE(15)
        SYMPA_SPD_PACK_CASE(16)
#undef SYMPA_SPD_PACK_CASE
        default: break;
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}
}  // namespace

int sympa_spd_table_pack(const double* table, int64_t num_rows, int n, void* pack, int64_t pack_bytes, int32_t* status,
                         void* stream) {
    return spd_pack_any(table, num_rows, n, pack, pack_bytes, status, nullptr, stream);
}

int sympa_spd_table_pack_refresh(const double* table, int64_t num_rows, int n, void* pack, int64_t pack_bytes, void* digest_state,
                                 int flags, int32_t* status, void* stream) {
    if (n < 6 || n > sympa::SPD_MAX_N) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd packed table: dims 6..16");
    if (table == nullptr || num_rows <= 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (digest_state == nullptr) return fail(SYMPA_ERR_BAD_ARG, "pack refresh: null digest state");
    const int rc = launch_table_digest(table, num_rows * (int64_t)(8 * n * n), digest_state, (flags & SYMPA_FLAG_DIGEST_FORCE) ? 1 : 0,
                                       reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) return rc;
    return spd_pack_any(table, num_rows, n, pack, pack_bytes, status,
                        reinterpret_cast<const unsigned*>(digest_state) + DIGEST_GUARD_WORD, stream);
}

int sympa_spd_model_forward_packed(const void* pack, int64_t pack_bytes, int64_t num_rows, int n, const int64_t* src,
                                   int64_t src_stride, const int64_t* dst, int64_t dst_stride, int64_t b, const double* scale,
                                   double scale_coef, double* out, int32_t* status, int flags, void* stream) {
    if (b > 0 && (src == nullptr || dst == nullptr)) return fail(SYMPA_ERR_BAD_ARG, "null index buffer");
    if (num_rows <= 0 && b > 0) return fail(SYMPA_ERR_BAD_ARG, "empty table");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    if (n < 6 || n > sympa::SPD_MAX_N) return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "spd packed forward: dims 6..16");
    if (pack == nullptr || pack_bytes < sympa_spd_table_pack_bytes(num_rows, n) || (reinterpret_cast<uintptr_t>(pack) & 15))
        return fail(SYMPA_ERR_BAD_ARG, "packed table: a 16-byte aligned buffer of sympa_spd_table_pack_bytes(num_rows, n) bytes");
    DistArgs a;
    std::memset(&a, 0, sizeof(a));
    a.base1 = reinterpret_cast<const double*>(pack);
    a.base2 = a.base1;
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
    a.flags = flags & ~(SYMPA_FLAG_GENERIC);
    return launch_spd(a, n, stream, true);
}

}  // extern "C"
