// The packed indexed forward of the upper model at dims 7, 8 as TWO kernels with TWO waves per SIMD each (round 5; C-ABI
// sympa_model_forward_packed with a workspace).
//
// Why.  One pair per lane, the n = 8 forward holds E (128 doubles) next to a factor and the accumulators: 412-512 registers, ONE
// wave per SIMD, and a lone wave issues an fp64 instruction every ~8.5 cycles.  tools/microbench/eigen_occupancy.hip measures what a
// second resident wave buys the eigenvalue stage (the same code at 1 / 2 / 3 waves per SIMD: 44.8 / 34.0 / 30.4 us per wave and
// SIMD slot): 1.33x.  A first split (one-pair-per-lane front, profiles/r05_packed_forward.txt block 1) lost that to its 512-register
// front kernel.  Here the FRONT runs two LANES per pair:
//   * E = A1 (Z2 - Z1) A2^T with REAL factors A_k (upper model) never mixes the planes: E_re = A1 D_re A2^T, E_im = A1 D_im A2^T.
//     Lane 2p works on the real plane of pair p, lane 2p + 1 on the imaginary plane -- 64 doubles of E per lane instead of 128, no
//     communication at all until the Gram matrix;
//   * H = E^H E:  Re H = E_re^T E_re + E_im^T E_im (each lane its own Gram matrix, one exchange-and-add), Im H[j][k] = sum_i (E_re[i][j]
//     E_im[i][k] - E_im[i][j] E_re[i][k]): each lane forms sum_i own[i][j] other[i][k] with the partner's columns arriving through
//     32-bit DPP moves (quad_perm [1,0,3,2]), one more exchange and a subtraction;
//   * the inverted factors are read from the LDS row by row as the products need them (the A tiles of the 32 pairs of a block are
//     10 KB), the triangles arrive through a ring of eight-row passes and are subtracted in place: ~230 registers, two waves per SIMD;
//   * H (n^2 doubles per pair) and the pair's status word go to a caller-owned workspace [tile of 64 pairs][entry][lane]; the
//     EIGEN kernel (one pair per lane, 170 registers, two waves per SIMD) reads it back: Householder + lockstep QL, log1p, metric.
#pragma once
#include "siegel_packed_kernel.hpp"

namespace sympa_hip {

template <int N>
struct SplitRow {
    using P = sympa::PointPack<N, sympa::MODEL_UPPER>;
    static constexpr int TRI = P::TRI;                       // doubles of one plane's triangle
    static constexpr int TC = TRI / 2;                       // ... in 16-byte chunks
    static constexpr int ZC = 2 * TC;                        // chunks of both triangles: what a Z pass fetches of a row
    static constexpr int ALEN = N + P::LOW;                  // doubles of the inverted factor (diagonal, strict lower part)
    static constexpr int AC = ALEN / 2;
    static constexpr int ROW_DOUBLES = PackRow<N, sympa::MODEL_UPPER>::ROW_DOUBLES;
    static constexpr int ZPITCH = ZC | 1;                    // LDS slots per row, odd
    static constexpr int ZROWS = 8;                          // pairs per Z pass
    static constexpr int ZBUF = ZROWS * ZPITCH;
    static constexpr int APITCH = AC | 1;
    static constexpr int A_PER_INSTR = 64 / APITCH;          // factor rows per DMA instruction (3 at n = 8)
    static constexpr int A_INSTR = (32 + A_PER_INSTR - 1) / A_PER_INSTR;
    static constexpr int A_INSTR_SLOTS = A_PER_INSTR * APITCH;
    static constexpr int ATILE = A_INSTR * A_INSTR_SLOTS;
    static constexpr int R0 = (2 * ZBUF > ATILE) ? 2 * ZBUF : ATILE;
    static constexpr int HLEN = N * N;                       // Herm<N>: d[N], then (re, im) of the strict upper part
    static constexpr int WS_ENTRIES = HLEN + 1;              // + the pair's status word
    static_assert(TRI % 2 == 0 && ALEN % 2 == 0, "chunk-aligned planes: dims 7, 8");
    static_assert(A_PER_INSTR >= 2 && A_PER_INSTR <= 4, "factor rows per DMA instruction");
};

constexpr bool packed_split_dims_ok(int n) { return n == 7 || n == 8; }

// swap a double with the other lane of my pair (lanes 2p, 2p + 1): two 32-bit DPP moves
__device__ __forceinline__ double pair_swap(const double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0xB1, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// the factor rows of the 32 pairs of my block: A_PER_INSTR rows per DMA instruction, row of pair q at
// tile + (q / A_PER_INSTR) * A_INSTR_SLOTS + (q % A_PER_INSTR) * APITCH
template <int N>
__device__ __forceinline__ void split_issue_factors(const double* __restrict__ pack, const int row, v2d* __restrict__ tile) {
    using S = SplitRow<N>;
    const int lane = threadIdx.x & 63;
    const int sub = lane / S::APITCH;                       // which of the rows of an instruction I fetch for
    const int c = lane - sub * S::APITCH;                   // chunk of that row (c == AC: the padding slot, nothing fetched)
    const bool on = sub < S::A_PER_INSTR && c < S::AC;
#pragma unroll
    for (int j = 0; j < S::A_INSTR; ++j) {
        // pair q = A_PER_INSTR j + sub; its row index sits in lanes 2q, 2q + 1
        int rr = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 0) & 31));
        if (S::A_PER_INSTR > 1) { const int r1 = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 1) & 31)); rr = sub == 1 ? r1 : rr; }
        if (S::A_PER_INSTR > 2) { const int r2 = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 2) & 31)); rr = sub == 2 ? r2 : rr; }
        if (S::A_PER_INSTR > 3) { const int r3 = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 3) & 31)); rr = sub == 3 ? r3 : rr; }
        const double* src = pack + (int64_t)rr * S::ROW_DOUBLES + 2 * (S::ZC + c);
        if (on && S::A_PER_INSTR * j + sub < 32)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(tile + j * S::A_INSTR_SLOTS), 16, 0, 0);
    }
}

// one Z pass: the two triangles (chunks [0, ZC)) of the rows of pairs 8 pass .. 8 pass + 7, one DMA instruction per row
template <int N>
__device__ __forceinline__ void split_issue_zpass(const double* __restrict__ pack, const int row, const int pass, v2d* __restrict__ buf) {
    using S = SplitRow<N>;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < S::ZROWS; ++j) {
        const int rr = __builtin_amdgcn_readlane(row, 2 * (S::ZROWS * pass + j));
        const double* src = pack + (int64_t)rr * S::ROW_DOUBLES + 2 * lane;
        if (lane < S::ZC) __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(buf + j * S::ZPITCH), 16, 0, 0);
    }
}

// my plane's triangle of the rows of a pass: SECOND = false: dh = it (the pair's second point); true: dh -= it (the first)
template <int N, bool SECOND>
__device__ __forceinline__ void split_read_zpass(const v2d* __restrict__ buf, const int pass, double (&dh)[SplitRow<N>::TRI]) {
    using S = SplitRow<N>;
    const int lane = threadIdx.x & 63;
    const int pr = lane >> 1, part = lane & 1;
    if ((pr >> 3) == pass) {
        const v2d* mine = buf + (pr & 7) * S::ZPITCH + part * S::TC;
        constexpr int G = 9;
#pragma unroll
        for (int c0 = 0; c0 < S::TC; c0 += G) {
            v2d q[G];
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (c0 + g < S::TC) q[g] = mine[c0 + g];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int c = c0 + g;
                if (c < S::TC) {
                    if (SECOND) { dh[2 * c] -= q[g].x; dh[2 * c + 1] -= q[g].y; }
                    else { dh[2 * c] = q[g].x; dh[2 * c + 1] = q[g].y; }
                }
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(64, 2) void packed_front2_kernel(const PackedArgs a) {
    using S = SplitRow<N>;
    using P = typename S::P;
    __shared__ v2d r0[S::R0];
    __shared__ v2d r1[S::ATILE];
    const int lane = threadIdx.x & 63;
    const int pr = lane >> 1, part = lane & 1;
    const unsigned ht = blockIdx.x;                 // half-tile: 32 pairs; tiles of 64 pairs are numbered through the batches
    const unsigned t = ht >> 1;
    const int kb = packed_batch_of(a, t);
    const unsigned t0 = (kb == 0) ? 0u : a.tile_end[kb - 1];
    const int64_t i = (int64_t)(t - t0) * 64 + (ht & 1u) * 32 + pr;
    const int64_t ii = i < a.b[kb] ? i : a.b[kb] - 1;
    int64_t x1 = __builtin_nontemporal_load(a.idx1[kb] + ii * a.stride1);
    int64_t x2 = __builtin_nontemporal_load(a.idx2[kb] + ii * a.stride2);
    int row1, row2, st;
    packed_ids_check(a, x1, x2, row1, row2, st);
    v2d* zb0 = r0;
    v2d* zb1 = r0 + S::ZBUF;
    split_issue_factors<N>(a.pack, row1, r1);                          // A1: in flight through the whole ring
    split_issue_zpass<N>(a.pack, row2, 0, zb0);                        // passes 0..3: the second point, 4..7: the first
    split_issue_zpass<N>(a.pack, row2, 1, zb1);
    double dh[S::TRI];
#pragma unroll
    for (int k = 0; k < S::TRI; ++k) dh[k] = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // pass s has landed when at most the ZROWS instructions of pass s + 1 are outstanding (the factor tile is older than all)
        if (s + 1 < 8) wait_vmcnt<S::ZROWS>();
        else wait_vmcnt<0>();
        wave_lds_fence();
        const v2d* cur = (s & 1) ? zb1 : zb0;
        if (s < 4) split_read_zpass<N, false>(cur, s, dh);
        else split_read_zpass<N, true>(cur, s - 4, dh);
        if (s + 2 < 8) {
            __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): the reads of the buffer being refilled are complete
            wave_lds_fence();
            v2d* nxt = (s & 1) ? zb1 : zb0;
            if (s + 2 < 4) split_issue_zpass<N>(a.pack, row2, s + 2, nxt);
            else split_issue_zpass<N>(a.pack, row1, s + 2 - 4, nxt);
        }
    }
    // the ring's region is free: A2 lands there while T = A1 D is formed from the tile that has long arrived
    __builtin_amdgcn_s_waitcnt(0xC07F);
    wave_lds_fence();
    split_issue_factors<N>(a.pack, row2, r0);
    const int aoff = (pr / S::A_PER_INSTR) * S::A_INSTR_SLOTS + (pr % S::A_PER_INSTR) * S::APITCH;
    // volatile: every factor entry is read where its product needs it (hoisted to the top, the 36 doubles of a factor are 72
    // registers of a kernel that has to stay at 256)
    const volatile double* a1 = reinterpret_cast<const volatile double*>(r1 + aoff);     // [diag N | strict lower, row-major]
    const volatile double* a2 = reinterpret_cast<const volatile double*>(r0 + aoff);
    double e[N][N];                                                     // my plane of D, then of T, then of E
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = r; c < N; ++c) { e[r][c] = dh[sympa::tri_index(N, r, c)]; e[c][r] = e[r][c]; }
    const bool ok1 = sympa::d_finite(a1[0]);
    // T = A1 D, rows N-1 .. 0 (row r needs rows k <= r of D only)
#pragma unroll
    for (int rr = 0; rr < N; ++rr) {
        const int r = N - 1 - rr;
        const double dg = a1[r];
        double lr[N];
#pragma unroll
        for (int k = 0; k < r; ++k) lr[k] = a1[N + sympa::low_index(r, k)];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            double x = dg * e[r][c];
#pragma unroll
            for (int k = 0; k < r; ++k) x = sympa::d_fma(lr[k], e[k][c], x);
            e[r][c] = x;
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) asm volatile("" : "+v"(e[r][c]));       // (T complete before E starts: see below)
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0): A2 has landed
    wave_lds_fence();
    const bool ok = ok1 && sympa::d_finite(a2[0]);
    // E = T A2^T, columns N-1 .. 0
#pragma unroll
    for (int cc = 0; cc < N; ++cc) {
        const int c = N - 1 - cc;
        const double dg = a2[c];
        double lr[N];
#pragma unroll
        for (int k = 0; k < c; ++k) lr[k] = a2[N + sympa::low_index(c, k)];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            double x = e[r][c] * dg;
#pragma unroll
            for (int k = 0; k < c; ++k) x = sympa::d_fma(e[r][k], lr[k], x);
            e[r][c] = x;
        }
    }
    // pin E here: everything above is ONE basic block of ~3 000 instructions, and left to itself the instruction selection interleaves
    // the products with the Gram sums below (every value of T, E and the partial sums alive at once: 100+ registers in scratch)
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) asm volatile("" : "+v"(e[r][c]));
    // ---- H = E^H E.  The workspace slot of my pair: entry e of tile t at ws + (t WS_ENTRIES + e) 64 + (ht & 1) 32 + pr
    // entry base (wave-uniform: a scalar register pair per store) + ONE 32-bit lane offset -- per-lane 64-bit addresses of 65 entries
    // would be 130 registers
    double* const wt = a.ws + ((int64_t)t * S::WS_ENTRIES) * 64 + (ht & 1u) * 32;
    const unsigned lo = (unsigned)pr * 8u;
    auto put = [&](const int entry, const double v) {
        __builtin_nontemporal_store(v, reinterpret_cast<double*>(reinterpret_cast<char*>(wt + (int64_t)entry * 64) + lo));
    };
    {
        // Re H = my Gram matrix + the partner's: formed, exchanged, added and stored (by the real-plane lane) entry by entry
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int k = j; k < N; ++k) {
                if (k == j) __builtin_amdgcn_sched_barrier(0);
                double g = 0.0;
#pragma unroll
                for (int r = 0; r < N; ++r) g = sympa::d_fma(e[r][j], e[r][k], g);
                g += pair_swap(g);
                if (j == 0 && k == 0 && (st & sympa::ST_BAD_INDEX)) g = __builtin_nan("");
                // Herm<N> order: d[N], then (re, im) of (j, k), j < k, row by row: pair index j (N - 1) - j (j - 1) / 2 + (k - j - 1)
                if (part == 0) put(j == k ? j : N + 2 * (j * N - j * (j + 1) / 2 + (k - j - 1)), g);
            }
    }
    {
        // Im H[j][k] = sum_r (E_re[r][j] E_im[r][k] - E_im[r][j] E_re[r][k]): both lanes form  x = sum_r own[r][j] other[r][k]  with the
        // partner's column k arriving through DPP; the imaginary-plane lane keeps  partner's x - its own x  and stores it
#pragma unroll
        for (int k = 1; k < N; ++k) {
            __builtin_amdgcn_sched_barrier(0);         // one partner column at a time (all seven fetched ahead are 112 registers)
            double oc[N];
#pragma unroll
            for (int r = 0; r < N; ++r) oc[r] = pair_swap(e[r][k]);
#pragma unroll
            for (int j = 0; j < k; ++j) {
                double x = 0.0;
#pragma unroll
                for (int r = 0; r < N; ++r) x = sympa::d_fma(e[r][j], oc[r], x);
                const double him = pair_swap(x) - x;                   // in the imaginary-plane lane: C1 - C2
                if (part == 1) put(N + 2 * (j * N - j * (j + 1) / 2 + (k - j - 1)) + 1, him);
            }
        }
    }
    if (part == 0) put(S::HLEN, (double)(st | (ok ? 0 : sympa::ST_NOT_PD)));
}

// eigen stage: H back from the workspace, one pair per lane, two waves per SIMD
template <int N>
__global__ __launch_bounds__(64, 2) void packed_eigen_kernel(const PackedArgs a) {
    using S = SplitRow<N>;
    const unsigned t = blockIdx.x;
    const int k = packed_batch_of(a, t);
    const unsigned t0 = (k == 0) ? 0u : a.tile_end[k - 1];
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)(t - t0) * 64 + lane;
    const bool live = i < a.b[k];
    const double* w = a.ws + ((int64_t)t * S::WS_ENTRIES) * 64 + lane;
    sympa::Herm<N> h;
#pragma unroll
    for (int j = 0; j < N; ++j) h.d[j] = __builtin_nontemporal_load(w + (int64_t)j * 64);
    {
        int e_ = N;
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int q = j + 1; q < N; ++q) {
                h.re[j][q] = __builtin_nontemporal_load(w + (int64_t)e_ * 64);
                h.im[j][q] = __builtin_nontemporal_load(w + (int64_t)(e_ + 1) * 64);
                e_ += 2;
            }
    }
    int st = (int)__builtin_nontemporal_load(w + (int64_t)S::HLEN * 64);
    double d = sympa::distance_from_h<N, sympa::MODEL_UPPER>(h, true, a.metric, a.metric_w, a.inv_eps, nullptr, st);
    if (st & sympa::ST_BAD_INDEX) d = __builtin_nan("");
    if (a.scale != nullptr) d *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);   // model.py:40-41
    if (live) __builtin_nontemporal_store(d, a.out[k] + i);
    if (a.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {
            if (flagged) atomicOr(&a.status[0], st);
            if (lane == 0) atomicAdd(&a.status[1], (int)__popcll(m));
        }
    }
}

template <int N>
int launch_packed_split(const PackedArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((packed_front2_kernel<N>), dim3(2 * a.tiles), dim3(64), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    hipLaunchKernelGGL((packed_eigen_kernel<N>), dim3(a.tiles), dim3(64), 0, s, a);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace sympa_hip
