// Indexed forward over a PACKED table, dims 5..8 (C-ABI sympa_table_pack / sympa_model_forward_packed): what Model.forward /
// forward_batches / evaluate run while the table does not change between batches (sympa/model.py:16-30, sympa/embeddings.py:29-34,
// sympa/runner.py:124-135,142-154: evaluation and the mAP matrix issue many batches over one table).
//
// (1) pack kernel, once per table version: every point is factored ONCE (Y = L L^T / I - W W^H = C C^H), the factor inverted, and
//     the point's upper triangles + the inverted factor stored as ONE contiguous row of the packed table (PointPack of
//     siegel_math.hpp: 108 doubles = 864 B at n = 8 upper where the reference row is 1 024 B).
// (2) pair kernel, one pair per lane, one wave per block, PERSISTENT waves (grid = what the chip holds at once, each wave walks
//     tiles of 64 pairs t, t + G, ...):
//       * the packed rows arrive through the LDS-DMA ring of siegel_gather.hpp, one coalesced instruction per row, the row index
//         through v_readlane (scalar base address);
//       * the SECOND point of a pair is read first; the triangles of the first one are subtracted from it IN PLACE as its chunks
//         come out of the LDS (D = Z2 - Z1), only its inverted factor is kept: a lane never holds two points' triangles
//         (144 doubles at n = 8 where the dense kernel holds 256 + two factors) -- no scratch in any instantiation, where the dense
//         bounded n = 8 kernel spills 652 bytes;
//       * E = A1 D A2^T by two triangular products in place -- no Cholesky, no division, no square root --, then H = E^H E,
//         eigenvalues, log, metric as in the dense kernel (siegel_math.hpp distance_from_h);
//       * the ids of the next tile are loaded one tile ahead and its first two passes are issued BEFORE the current tile's
//         arithmetic: the head of a tile (index load, first pass -- latency a lone wave cannot hide) disappears behind it.
//     Measured (profiles/r05_packed_forward.txt, 262 144 pairs, 45 500 rows, same box): upper n = 8 135.4 -> 126.9 us, n = 7
//     119.3 -> 103.6, n = 6 84.8 -> 67.7, n = 5 58.8 -> 53.4; bounded n = 8 245.2 -> 163.3, n = 7 152.5 -> 118.5.
//     A TWO-kernel form (front: gather + E + H to a workspace; eigen stage from the workspace at 170 registers = two waves per
//     SIMD) was built first and measured slower everywhere (upper n = 8: 169 us = 103 + 66): the 2 x 136 MB of H through the
//     workspace cost what the second wave brings.  It is not in the library; the measurements are in the profile.
#pragma once
#include "siegel_common.hpp"

// x 64 cycles per step of the staggered first round (measured on the final build of round 5, 262 144 pairs of 45 500 rows: n = 8
// 0 / 6 / 12 / 16 / 20 / 28 / 40 -> 129.7 / 124.1 / 117.7 / 116.9 / 116.6 / 121.9 / 131.4 us, n = 7 0 / 12 / 16 / 20 / 28 -> 99.0 / 90.3 /
// 90.3 / 93.5 / 99.5 us; sleeping AFTER the first tile's head is in flight: no better)
template <int N>
constexpr int packed_stagger_sleep() { return N >= 8 ? 20 : 16; }

namespace sympa_hip {

template <int N, int MODEL>
struct PackRow {
    using P = sympa::PointPack<N, MODEL>;
    static constexpr int LEN = P::LEN;
    static constexpr int K = (LEN + 1) / 2;                  // 16-byte chunks per packed row
    static constexpr int ROW_DOUBLES = 2 * K;                // row stride of the packed table
    static constexpr int PITCH = (K % 2 == 1) ? K : K + 1;   // LDS slots per row: odd, so ds_read_b128 is conflict-free
    static constexpr int ROWS = 16;                          // rows per pass
    static constexpr int IPP = ROWS * (K > 64 ? 2 : 1);      // LDS-DMA instructions per pass
    static constexpr int BUF_SLOTS = ROWS * PITCH;
    static constexpr int NBUF = 2;
    static constexpr int WAVE_SLOTS = NBUF * BUF_SLOTS;
    static_assert(K <= 128, "a packed row is at most two DMA instructions");
    static_assert(IPP <= 32, "vmcnt immediate");
};

constexpr int packed_dims_ok(int n) { return n >= 5 && n <= 8; }

// ---- (1) pack ---------------------------------------------------------------------------------------------------------------
template <int N, int MODEL>
__global__ __launch_bounds__(64) void table_pack_kernel(const double* __restrict__ table, const int64_t num_rows,
                                                        double* __restrict__ pack, int32_t* status, const unsigned* guard) {
    using R = PackRow<N, MODEL>;
    // sympa_table_pack_refresh: the digest kernel in front (table_digest.hpp) left 0 here when the table's bytes are those the
    // pack was made from -- nothing to do (wave-uniform scalar load)
    if (guard != nullptr && __builtin_nontemporal_load(guard) == 0u) return;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t ii = i < num_rows ? i : num_rows - 1;
    sympa::CMat<N> z;
    sympa::load_point<N>(table + ii * (2 * N * N), z);
    double p[R::ROW_DOUBLES];
    double q[R::LEN];
    const bool ok = sympa::pack_point<N, MODEL>(z, q);
#pragma unroll
    for (int k = 0; k < R::LEN; ++k) p[k] = q[k];
    if constexpr (R::ROW_DOUBLES > R::LEN) p[R::LEN] = 0.0;
    // a point outside the manifold has no factor: its inverted diagonal is stored as NaN, every pair it enters is flagged
    // SYMPA_ST_NOT_PD by the front kernel and comes out NaN (the dense path reports the same bit)
    if (!ok) {
#pragma unroll
        for (int k = 0; k < N; ++k) p[R::P::OFF_DIAG + k] = __builtin_nan("");
    }
    if (i < num_rows) {
        v2d* dst = reinterpret_cast<v2d*>(pack + i * R::ROW_DOUBLES);
#pragma unroll
        for (int c = 0; c < R::K; ++c) {
            v2d v;
            v.x = p[2 * c];
            v.y = p[2 * c + 1];
            dst[c] = v;
        }
    }
    if (status != nullptr) {
        const int bad = (i < num_rows && !ok) ? 1 : 0;
        const unsigned long long m = __ballot(bad);
        if (m != 0ull && (threadIdx.x & 63) == 0) {
            atomicOr(&status[0], sympa::ST_NOT_PD);
            atomicAdd(&status[1], (int)__popcll(m));
        }
    }
}

// ---- (2) pairs ----------------------------------------------------------------------------------------------------------------

// batch of tile t: binary search over <= 32 prefix ends (wave-uniform, scalar)
__device__ __forceinline__ int packed_batch_of(const PackedArgs& a, const unsigned t) {
    int lo = 0, hi = a.num_batches - 1;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int mid = (lo + hi) >> 1;
        const bool right = t >= a.tile_end[mid];
        lo = right ? mid + 1 : lo;
        hi = right ? hi : mid;
    }
    return lo;
}

// ids of this lane's pair of tile t (clamped for the idle tail lanes): the RAW loads only -- nothing here depends on the loaded
// values, so the wave does not wait for them until packed_ids_check runs, a whole tile later
__device__ __forceinline__ void packed_ids_load(const PackedArgs& a, const unsigned t, int64_t& x1, int64_t& x2) {
    const int k = packed_batch_of(a, t);
    const unsigned t0 = (k == 0) ? 0u : a.tile_end[k - 1];
    const int64_t i = (int64_t)(t - t0) * 64 + (threadIdx.x & 63);
    const int64_t ii = i < a.b[k] ? i : a.b[k] - 1;
    if (a.identity) {
        x1 = ii;
        x2 = ii;
        return;
    }
    x1 = __builtin_nontemporal_load(a.idx1[k] + ii * a.stride1);
    x2 = __builtin_nontemporal_load(a.idx2[k] + ii * a.stride2);
}
// out-of-range ids are flagged and replaced by row 0 (the reference raises IndexError)
__device__ __forceinline__ void packed_ids_check(const PackedArgs& a, int64_t x1, int64_t x2, int& r1, int& r2, int& st) {
    st = 0;
    if (x1 < 0 || x1 >= a.num_rows || x2 < 0 || x2 >= a.num_rows) {
        st = sympa::ST_BAD_INDEX;
        x1 = 0;
        x2 = 0;
    }
    r1 = (int)x1;
    r2 = (int)x2;
}

// One pass of the ring: sixteen rows of K 16-byte chunks each (row stride ROW_DOUBLES doubles), one LDS-DMA instruction per row (two
// when K > 64), row j of the pass at buf + j * PITCH.  The row index comes from v_readlane: a scalar base address.
template <int K, int ROW_DOUBLES, int PITCH>
__device__ __forceinline__ void ring_pass_issue(const double* __restrict__ base, const int row, const int pass, v2d* __restrict__ buf) {
    const int lane = threadIdx.x & 63;
    // (lds_dest: no null check per instruction; ONE lane mask around the sixteen instructions of the pass instead of one save / restore
    // of EXEC each.  The sixteen row indices are read BEFORE the mask: v_readlane itself does not look at EXEC, but behind the branch
    // the compiler is free to compute `row` for the active lanes only -- lanes K..63 then hold garbage, and a pass reads them.
    // Dims 8 upper: 1 792 -> ~950 scalar instructions per tile)
    int rr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) rr[j] = __builtin_amdgcn_readlane(row, 16 * pass + j);
    if (K >= 64 || lane < K) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double* src = base + (int64_t)rr[j] * ROW_DOUBLES + 2 * lane;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, lds_dest(buf + j * PITCH), 16, 0, 0);
        }
    }
    if constexpr (K > 64) {
        if (lane < K - 64) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double* src = base + (int64_t)rr[j] * ROW_DOUBLES + 2 * lane;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + 128), lds_dest(buf + j * PITCH + 64), 16, 0, 0);
            }
        }
    }
}

template <int N, int MODEL>
__device__ __forceinline__ void packed_pass_issue(const double* __restrict__ base, const int row, const int pass,
                                                  v2d* __restrict__ buf) {
    using R = PackRow<N, MODEL>;
    ring_pass_issue<R::K, R::ROW_DOUBLES, R::PITCH>(base, row, pass, buf);
}

// The rows of a pass back into registers (the sixteen lanes of the pass only).  SECOND = false: the row of the pair's SECOND point
// (dst) lands in p2 whole.  SECOND = true, the FIRST point (src): its triangles are subtracted from p2's in place (D = Z2 - Z1), only
// its inverted factor is kept (p1a) -- a lane never holds two points' triangles, 144 doubles at n = 8 instead of 216.
template <int N, int MODEL, bool SECOND>
__device__ __forceinline__ void packed_pass_read(const v2d* __restrict__ buf, const int pass, double (&p2)[PackRow<N, MODEL>::ROW_DOUBLES],
                                                 double (&p1a)[PackRow<N, MODEL>::ROW_DOUBLES - 2 * PackRow<N, MODEL>::P::TRI]) {
    using R = PackRow<N, MODEL>;
    constexpr int ZLEN = 2 * R::P::TRI;
    const int lane = threadIdx.x & 63;
    if ((lane >> 4) == pass) {
        const v2d* mine = buf + (lane & 15) * R::PITCH;
        if constexpr (!SECOND) {
#pragma unroll
            for (int c = 0; c < R::K; ++c) {
                const v2d q = mine[c];
                p2[2 * c] = q.x;
                p2[2 * c + 1] = q.y;
            }
        } else {
            // groups of G chunks: G reads in flight, then their subtractions (one read, one dependent subtraction at a time would
            // expose the LDS latency 54 times per pass -- this unit keeps the source order)
            constexpr int G = 12;
#pragma unroll
            for (int c0 = 0; c0 < R::K; c0 += G) {
                v2d q[G];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (c0 + g < R::K) q[g] = mine[c0 + g];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int c = c0 + g;
                    if (c < R::K) {
                        if (2 * c < ZLEN) p2[2 * c] -= q[g].x; else p1a[2 * c - ZLEN] = q[g].x;
                        if (2 * c + 1 < ZLEN) p2[2 * c + 1] -= q[g].y; else p1a[2 * c + 1 - ZLEN] = q[g].y;
                    }
                }
            }
        }
    }
}

// p1a as the first operand of e_from_packed<DIFF>: only the factor part is ever read
template <int N, int MODEL>
struct FactorOnly {
    const double* a;
    __device__ __forceinline__ double operator[](int k) const { return a[k - 2 * sympa::PointPack<N, MODEL>::TRI]; }
};

template <int N, int MODEL>
__global__ __launch_bounds__(64, 1) void packed_forward_kernel(const PackedArgs a) {
    using R = PackRow<N, MODEL>;
    using P = sympa::PointPack<N, MODEL>;
    __shared__ v2d tile[R::WAVE_SLOTS];
    v2d* buf0 = tile;
    v2d* buf1 = tile + R::BUF_SLOTS;
    // staggered first round, as in siegel_dist_kernel.hpp (tables beyond the L2s: CU j of every XCD starts j x 0.5 us late, the
    // waves stay out of step from there on; upper n = 8: 135.9 -> 126.9 us)
    if (a.stagger && blockIdx.x < 1024u) {
        const int k = (int)((blockIdx.x >> 5) & 31u);
        for (int j = 0; j < k; ++j) __builtin_amdgcn_s_sleep(packed_stagger_sleep<N>());
    }
    unsigned t = blockIdx.x;
    if (t >= a.tiles) return;
    int r1, r2, st;
    int64_t x1, x2;
    packed_ids_load(a, t, x1, x2);
    packed_ids_check(a, x1, x2, r1, r2, st);
    packed_pass_issue<N, MODEL>(a.pack, r2, 0, buf0);        // passes 0..3: the pair's second point, 4..7: the first
    packed_pass_issue<N, MODEL>(a.pack, r2, 1, buf1);
    // the ids of the NEXT tile are always one tile ahead of the passes that need them (loaded behind the previous prefetch)
    unsigned tn = t + gridDim.x;
    bool more = tn < a.tiles;                      // wave-uniform
    x1 = 0;
    x2 = 0;
    if (more) packed_ids_load(a, tn, x1, x2);
    for (;;) {
        // the ring below counts on the DMA passes of THIS tile being the only outstanding vector-memory operations
        __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0): passes 0, 1 (issued long ago), the next ids, the previous tile's stores
        double p2[R::ROW_DOUBLES], p1a[R::ROW_DOUBLES - 2 * P::TRI];
#pragma unroll
        for (int k = 0; k < R::ROW_DOUBLES; ++k) p2[k] = 0.0;
#pragma unroll
        for (int k = 0; k < R::ROW_DOUBLES - 2 * P::TRI; ++k) p1a[k] = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // pass s has landed: s = 0 by the vmcnt(0) above; then at most the IPP instructions of pass s + 1 are outstanding
            if (s > 0) {
                if (s + 1 < 8) wait_vmcnt<R::IPP>();
                else wait_vmcnt<0>();
            }
            wave_lds_fence();
            const v2d* cur = (s & 1) ? buf1 : buf0;
            if (s < 4) packed_pass_read<N, MODEL, false>(cur, s, p2, p1a);
            else packed_pass_read<N, MODEL, true>(cur, s - 4, p2, p1a);
            if (s + 2 < 8) {
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the reads of the buffer being refilled are complete
                wave_lds_fence();
                v2d* nxt = (s & 1) ? buf1 : buf0;
                if (s + 2 < 4) packed_pass_issue<N, MODEL>(a.pack, r2, s + 2, nxt);
                else packed_pass_issue<N, MODEL>(a.pack, r1, s + 2 - 4, nxt);
            }
        }
        // the head of the next tile goes out before this tile's arithmetic, the ids of the one after that behind it
        const unsigned tnn = tn + gridDim.x;
        const bool more2 = more && tnn < a.tiles;
        int n1 = 0, n2 = 0, nst = 0;
        if (more) packed_ids_check(a, x1, x2, n1, n2, nst);     // (loaded a tile ago)
        if (more) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            wave_lds_fence();
            packed_pass_issue<N, MODEL>(a.pack, n2, 0, buf0);
            packed_pass_issue<N, MODEL>(a.pack, n2, 1, buf1);
        }
        if (more2) packed_ids_load(a, tnn, x1, x2);
        // ---- arithmetic of tile t
        const bool ok = sympa::d_finite(p1a[P::OFF_DIAG - 2 * P::TRI]) && sympa::d_finite(p2[P::OFF_DIAG]);
        sympa::Herm<N> h;
        {
            sympa::CMat<N> e;
            const FactorOnly<N, MODEL> p1{p1a};
            sympa::e_from_packed<N, MODEL, true>(p1, p2, e);
            sympa::gram<N>(e, h);
        }
        int flags = st | (ok ? 0 : sympa::ST_NOT_PD);
        const int kb = packed_batch_of(a, t);
        const unsigned t0 = (kb == 0) ? 0u : a.tile_end[kb - 1];
        const int64_t i = (int64_t)(t - t0) * 64 + (threadIdx.x & 63);
        const bool live = i < a.b[kb];
        double d = sympa::distance_from_h<N, MODEL>(h, true, a.metric, a.metric_w, a.inv_eps, nullptr, flags);
        if (flags & sympa::ST_BAD_INDEX) d = __builtin_nan("");
        if (a.scale != nullptr) d *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);   // model.py:40-41
        if (live) __builtin_nontemporal_store(d, a.out[kb] + i);
        if (a.status != nullptr) {
            const int flagged = (live && flags != 0) ? 1 : 0;
            const unsigned long long m = __ballot(flagged);
            if (m != 0ull) {
                if (flagged) atomicOr(&a.status[0], flags);
                if ((threadIdx.x & 63) == 0) atomicAdd(&a.status[1], (int)__popcll(m));
            }
        }
        if (!more) break;
        t = tn; r1 = n1; r2 = n2; st = nst;
        tn = tnn; more = more2;
    }
}

// ---- the same structure over the reference's DENSE rows, upper model (round 5): persistent waves, prefetched tile head, the
// first point's Re triangle subtracted from the second's in place -- what sympa_model_forward / sympa_siegel_dist_fwd run at dims
// 7, 8 when no packed table is at hand (a table that changes before every call, training's forward, pre-gathered points).  Same
// arithmetic as siegel_dist_kernel (Cholesky of both imaginary parts, two triangular solves, distance_from_h), same values.
// Measured against it (profiles/r05_dense_persistent_ab.txt, 262 144 pairs of 45 500 rows): n = 8 144.9 -> 135.0 us, n = 7 118.1 ->
// 110.3; n = 6 84.4 -> 88.6 and n = 5 57.4 -> 58.3 (their one-launch kernels already hold two waves per SIMD): dims 7, 8 only.
template <int N>
struct DenseRow {
    static constexpr int K = N * N;                           // 16-byte chunks per [2, N, N] row
    static constexpr int ROW_DOUBLES = 2 * N * N;
    static constexpr int PITCH = (K % 2 == 1) ? K : K + 1;
    static constexpr int BUF_SLOTS = 16 * PITCH;
    static constexpr int WAVE_SLOTS = 2 * BUF_SLOTS;
    static constexpr int TRI = N * (N + 1) / 2;
    // element `half` of chunk c: plane, (i, j); kept when i <= j
    static constexpr bool keep(int c, int half) {
        const int f = 2 * c + half, g = f % (N * N);
        return (g / N) <= (g % N);
    }
    static constexpr bool chunk_needed(int c) { return keep(c, 0) || keep(c, 1); }
    static constexpr int plane(int c, int half) { return (2 * c + half) / (N * N); }
    static constexpr int tri_of(int c, int half) {
        const int g = (2 * c + half) % (N * N);
        return sympa::tri_index(N, g / N, g % N);
    }
};

// SECOND = false: the pair's second point: Re triangle -> xr, Im triangle -> y2.  SECOND = true, the first point: xr -= its Re
// triangle (D = Z2 - Z1, real part), its Im triangle -> y1.  Chunks that hold no upper-triangle element are not read.
template <int N, bool SECOND>
__device__ __forceinline__ void dense_pass_read(const v2d* __restrict__ buf, const int pass, double (&xr)[DenseRow<N>::TRI],
                                                double (&ya)[DenseRow<N>::TRI]) {
    using R = DenseRow<N>;
    const int lane = threadIdx.x & 63;
    if ((lane >> 4) == pass) {
        const v2d* mine = buf + (lane & 15) * R::PITCH;
        constexpr int G = 12;
#pragma unroll
        for (int c0 = 0; c0 < R::K; c0 += G) {
            v2d q[G];
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (c0 + g < R::K && R::chunk_needed(c0 + g)) q[g] = mine[c0 + g];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int c = c0 + g;
                if (c < R::K && R::chunk_needed(c)) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (R::keep(c, h)) {
                            const double v = h ? q[g].y : q[g].x;
                            if (R::plane(c, h) == 0) {
                                if (SECOND) xr[R::tri_of(c, h)] -= v; else xr[R::tri_of(c, h)] = v;
                            } else {
                                ya[R::tri_of(c, h)] = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(64, 1) void dense_forward_kernel(const PackedArgs a) {
    using R = DenseRow<N>;
    constexpr int TRI = R::TRI;
    __shared__ v2d tile[R::WAVE_SLOTS];
    v2d* buf0 = tile;
    v2d* buf1 = tile + R::BUF_SLOTS;
    if (a.stagger && blockIdx.x < 1024u) {
        const int k = (int)((blockIdx.x >> 5) & 31u);
        for (int j = 0; j < k; ++j) __builtin_amdgcn_s_sleep(packed_stagger_sleep<N>());
    }
    unsigned t = blockIdx.x;
    if (t >= a.tiles) return;
    int r1, r2, st;
    int64_t x1, x2;
    packed_ids_load(a, t, x1, x2);
    packed_ids_check(a, x1, x2, r1, r2, st);
    ring_pass_issue<R::K, R::ROW_DOUBLES, R::PITCH>(a.base2, r2, 0, buf0);      // passes 0..3: the second point, 4..7: the first
    ring_pass_issue<R::K, R::ROW_DOUBLES, R::PITCH>(a.base2, r2, 1, buf1);
    unsigned tn = t + gridDim.x;
    bool more = tn < a.tiles;
    x1 = 0;
    x2 = 0;
    if (more) packed_ids_load(a, tn, x1, x2);
    for (;;) {
        __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0): passes 0, 1, the next ids, the previous tile's store
        double dre[TRI], y2[TRI], y1[TRI];
#pragma unroll
        for (int k = 0; k < TRI; ++k) { dre[k] = 0.0; y2[k] = 0.0; y1[k] = 0.0; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s > 0) {
                if (s + 1 < 8) wait_vmcnt<16>();
                else wait_vmcnt<0>();
            }
            wave_lds_fence();
            const v2d* cur = (s & 1) ? buf1 : buf0;
            if (s < 4) dense_pass_read<N, false>(cur, s, dre, y2);
            else dense_pass_read<N, true>(cur, s - 4, dre, y1);
            if (s + 2 < 8) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                wave_lds_fence();
                v2d* nxt = (s & 1) ? buf1 : buf0;
                if (s + 2 < 4) ring_pass_issue<R::K, R::ROW_DOUBLES, R::PITCH>(a.base2, r2, s + 2, nxt);
                else ring_pass_issue<R::K, R::ROW_DOUBLES, R::PITCH>(a.pack, r1, s + 2 - 4, nxt);
            }
        }
        const unsigned tnn = tn + gridDim.x;
        const bool more2 = more && tnn < a.tiles;
        int n1 = 0, n2 = 0, nst = 0;
        if (more) packed_ids_check(a, x1, x2, n1, n2, nst);
        if (more) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            wave_lds_fence();
            ring_pass_issue<R::K, R::ROW_DOUBLES, R::PITCH>(a.base2, n2, 0, buf0);
            ring_pass_issue<R::K, R::ROW_DOUBLES, R::PITCH>(a.base2, n2, 1, buf1);
        }
        if (more2) packed_ids_load(a, tnn, x1, x2);
        // ---- arithmetic of tile t: as pair_distance_mats<N, MODEL_UPPER>
        sympa::Herm<N> h;
        bool ok;
        {
            sympa::CMat<N> e;
            {
                double ym[N][N];
                sympa::Tri<N, false> l1, l2;
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = i; j < N; ++j) ym[i][j] = y1[sympa::tri_index(N, i, j)];
                ok = sympa::chol_real<N>(ym, l1);
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = i; j < N; ++j) ym[i][j] = y2[sympa::tri_index(N, i, j)];
                ok = sympa::chol_real<N>(ym, l2) && ok;
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = i; j < N; ++j) {
                        const double xr = dre[sympa::tri_index(N, i, j)];
                        const double xi = y2[sympa::tri_index(N, i, j)] - y1[sympa::tri_index(N, i, j)];
                        e.re[i][j] = xr; e.re[j][i] = xr;
                        e.im[i][j] = xi; e.im[j][i] = xi;
                    }
                sympa::solve_left<N, false>(l1, e);
                sympa::solve_right_t<N, false>(l2, e);
            }
            sympa::gram<N>(e, h);
        }
        int flags = st;
        const int kb = packed_batch_of(a, t);
        const unsigned t0 = (kb == 0) ? 0u : a.tile_end[kb - 1];
        const int64_t i = (int64_t)(t - t0) * 64 + (threadIdx.x & 63);
        const bool live = i < a.b[kb];
        double d = sympa::distance_from_h<N, sympa::MODEL_UPPER>(h, ok, a.metric, a.metric_w, a.inv_eps, nullptr, flags);
        if (flags & sympa::ST_BAD_INDEX) d = __builtin_nan("");
        if (a.scale != nullptr) d *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);   // model.py:40-41
        if (live) __builtin_nontemporal_store(d, a.out[kb] + i);
        if (a.status != nullptr) {
            const int flagged = (live && flags != 0) ? 1 : 0;
            const unsigned long long m = __ballot(flagged);
            if (m != 0ull) {
                if (flagged) atomicOr(&a.status[0], flags);
                if ((threadIdx.x & 63) == 0) atomicAdd(&a.status[1], (int)__popcll(m));
            }
        }
        if (!more) break;
        t = tn; r1 = n1; r2 = n2; st = nst;
        tn = tnn; more = more2;
    }
}

template <int N, int MODEL>
int launch_table_pack(const double* table, int64_t num_rows, double* pack, int32_t* status, const unsigned* guard, hipStream_t s) {
    hipLaunchKernelGGL((table_pack_kernel<N, MODEL>), dim3((unsigned)((num_rows + 63) / 64)), dim3(64), 0, s, table, num_rows, pack,
                       status, guard);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

// persistent grid: as many one-wave blocks as the chip holds at once (registers / LDS of the instantiation)
// (asked of the runtime ONCE per kernel instantiation and device: the query sits in front of a ~100 us kernel and inside graph
// captures otherwise -- round-5 advice)
template <class K>
unsigned resident_blocks(K kern, unsigned cus) {
    static int cached[16] = {0};          // per instantiation of this template, by device ordinal
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int per_cu = cached[dev];
    if (per_cu <= 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
        cached[dev] = per_cu;
    }
    return cus * (unsigned)per_cu;
}

template <int N, int MODEL>
int launch_packed_forward(const PackedArgs& a, unsigned cus, hipStream_t s) {
    const unsigned res = resident_blocks(packed_forward_kernel<N, MODEL>, cus);
    hipLaunchKernelGGL((packed_forward_kernel<N, MODEL>), dim3(a.tiles < res ? a.tiles : res), dim3(64), 0, s, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

template <int N>
int launch_dense_forward(const PackedArgs& a, unsigned cus, hipStream_t s) {
    const unsigned res = resident_blocks(dense_forward_kernel<N>, cus);
    hipLaunchKernelGGL((dense_forward_kernel<N>), dim3(a.tiles < res ? a.tiles : res), dim3(64), 0, s, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace sympa_hip
