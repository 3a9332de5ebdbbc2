// Device-only helpers shared by the product kernels (siegel_dist.hip) and the ablation tools
// (tools/microbench): cooperative, coalesced row gather through a wave-private LDS tile.
#pragma once
#include <hip/hip_runtime.h>
#include "siegel_math.hpp"

// ---------------------------------------------------------------------------------------------
// Row gather.  A lane-per-pair load of a 16 n^2-byte row touches 64 different cache lines per wave
// instruction and is bound by the texture-address unit (measured: 3 us of a 9.6 us launch at n = 4,
// profiles/r01_kernel_anatomy.txt).  For n <= 4 the wave therefore loads rows COOPERATIVELY: a row is
// C = n^2 chunks of 16 B, consecutive lanes fetch consecutive chunks (coalesced 16n^2-byte segments),
// the chunks are staged in a wave-private LDS tile [pair][SLOTS] and each lane then reads back its own
// row with ds_read_b128.  SLOTS (row pitch in 16-B slots) is odd, so the 16 lanes of a b128 read group
// fall on 16 distinct 4-bank slots: conflict-free.
// The tile belongs to ONE wave: LDS operations of a wave are processed in order, so no block barrier
// is needed, only a compiler-level fence that keeps the cross-lane write -> read order.
// ---------------------------------------------------------------------------------------------
typedef double v2d __attribute__((ext_vector_type(2)));   // one 16-B chunk

template <int N>
struct Tile {
    static constexpr int C = N * N;                       // 16-B chunks per row
    static constexpr int SLOTS = (C % 2 == 1) ? C : C + 1;
    static constexpr bool STAGED = (N >= 2 && N <= 4);
    static constexpr int WAVE_SLOTS = STAGED ? 64 * SLOTS : 1;
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Issues the coalesced loads of the 64 rows `row` (one per lane) of `base`; chunk t of this lane.
template <int N>
__device__ __forceinline__ void gather_issue(const double* __restrict__ base, const int row, v2d (&v)[N * N]) {
    constexpr int C = Tile<N>::C;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int g = t * 64 + lane;
        const int p = g / C, c = g - p * C;
        const int r = __shfl(row, p);
        v[t] = reinterpret_cast<const v2d*>(base + (int64_t)r * (2 * N * N))[c];
    }
}

// Transposes the staged chunks through the wave's tile: afterwards lane i holds row i.
template <int N>
__device__ __forceinline__ void gather_transpose(const v2d (&v)[N * N], v2d* __restrict__ tile, sympa::CMat<N>& z) {
    constexpr int C = Tile<N>::C, SLOTS = Tile<N>::SLOTS;
    const int lane = threadIdx.x & 63;
    wave_lds_fence();   // earlier reads of the tile (previous side) are complete in program order
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int g = t * 64 + lane;
        const int p = g / C, c = g - p * C;
        tile[p * SLOTS + c] = v[t];
    }
    wave_lds_fence();
    v2d q[C];
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = tile[lane * SLOTS + c];
    // flat index f of [2, n, n]: element f lives in chunk f / 2, half f % 2
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int fr = (i <= j) ? i * N + j : j * N + i;
            const int fi = N * N + fr;
            z.re[i][j] = (fr & 1) ? q[fr >> 1].y : q[fr >> 1].x;
            z.im[i][j] = (fi & 1) ? q[fi >> 1].y : q[fi >> 1].x;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (n = 2, 4: C = n^2 divides 64).  `global_load_lds_dwordx4` moves 64 x 16 B straight
// from the per-lane SOURCE addresses into 1 KiB of contiguous LDS (no VGPR staging, no ds_write).
// Instruction j of a side fetches the R = 64 / C rows of pairs {j, j + C, j + 2C, ...} (16 lanes = one
// coalesced row); its 1 KiB lands at j * 1040 B.  Lane i then finds chunk k of its row at
//      (i % C) * 1040 + (i / C) * 16 C + 16 k        [bytes]
// and the odd slot pitch 1040 / 16 = 65 makes every ds_read_b128 lane group conflict-free.
// ---------------------------------------------------------------------------------------------
template <int N>
struct DmaTile {
    static constexpr int C = N * N;
    static constexpr bool ENABLED = (N == 2 || N == 4);
    static constexpr int INSTR_SLOTS = 65;                  // 1040 B
    static constexpr int SIDE_SLOTS = ENABLED ? C * INSTR_SLOTS : 1;
    static constexpr int WAVE_SLOTS = 2 * SIDE_SLOTS;       // both endpoints in flight
    static constexpr int WAVE_SLOTS_LOW = SIDE_SLOTS;       // one endpoint at a time (half the LDS)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// The LDS destination of a DMA instruction from a (generic) pointer into a __shared__ array: its low 32 bits ARE the LDS offset.
// A plain cast to the LDS address space carries a null check -- s_cmp_lg_u64 / s_cselect per instruction, three scalar
// instructions in front of every global_load_lds (round 5: 1 792 -> 947 scalar instructions per tile of the dims-8 packed forward
// together with one lane mask per pass, profiles/r05_packed_forward.txt block 9).
__device__ __forceinline__ lds_ptr_t lds_dest(const void* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
    return (lds_ptr_t)(unsigned)reinterpret_cast<unsigned long long>(p);
#pragma clang diagnostic pop
}
// The same from a SCALAR offset: a wave's tile starts at (threadIdx.x >> 6) * WAVE_SLOTS -- wave-uniform, but a vector value to the
// compiler, so every DMA instruction paid a v_mad + v_readfirstlane for its M0.  lds_offset_uniform() says it once per gather.
__device__ __forceinline__ unsigned lds_offset_uniform(const void* p) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<unsigned long long>(p));
}
__device__ __forceinline__ lds_ptr_t lds_dest_at(const unsigned offset) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
    return (lds_ptr_t)offset;
#pragma clang diagnostic pop
}
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// (Masking the lanes of the 4 chunks per row that hold only lower-triangle entries off the DMA instruction
// was measured 3 % SLOWER than fetching whole rows -- tools/ab_bench.py, 7.93 vs 7.69 us -- so rows are
// fetched whole.)
// Row index of pair (j + C r) for the lanes of group r = lane / C: the source lane C r + j sits in the same
// group, so this is a broadcast inside a row of 16 lanes (C = 16: DPP row_newbcast:j) or inside a quad
// (C = 4: DPP quad_perm [j,j,j,j]) -- one VALU move instead of a ds_bpermute round trip through the LDS.
template <int C, int J>
__device__ __forceinline__ int group_bcast(const int v) {
    // (bound_ctrl = true: every lane has a valid source under these controls, so nothing changes in the result -- but the old value of
    // the destination no longer matters and the compiler stops initialising the register in front of every DPP move)
    if constexpr (C == 16) return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true);
    else if constexpr (C == 4) return __builtin_amdgcn_update_dpp(0, v, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, true);
    else return __shfl(v, J + C * ((int)(threadIdx.x & 63) / C));
}

template <int N, int J>
struct DmaIssue {
    static __device__ __forceinline__ void run(const double* __restrict__ base, const int row, const int c,
                                               const unsigned side) {
        constexpr int C = DmaTile<N>::C;
        const int rr = group_bcast<C, J>(row);
        // 32-bit byte offset from the (wave-uniform) table base: scalar base + vector offset addressing, no 64-bit
        // address arithmetic per instruction (tables of n <= 4 are limited to 4 GiB, checked on the host)
        const unsigned off = (unsigned)rr * (unsigned)(16 * N * N) + (unsigned)(16 * c);
        const char* src = reinterpret_cast<const char*>(base) + off;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, lds_dest_at(side + (unsigned)(J * DmaTile<N>::INSTR_SLOTS) * 16u), 16, 0, 0);
        if constexpr (J + 1 < C) DmaIssue<N, J + 1>::run(base, row, c, side);
    }
};

template <int N>
__device__ __forceinline__ void dma_issue(const double* __restrict__ base, const int row, v2d* __restrict__ side) {
    constexpr int C = DmaTile<N>::C;
    const int lane = threadIdx.x & 63;
    const int c = lane % C;
    DmaIssue<N, 0>::run(base, row, c, lds_offset_uniform(side));
}

template <int N>
__device__ __forceinline__ void dma_read(const v2d* __restrict__ side, sympa::CMat<N>& z) {
    constexpr int C = DmaTile<N>::C;
    const int lane = threadIdx.x & 63;
    const v2d* mine = side + (lane % C) * DmaTile<N>::INSTR_SLOTS + (lane / C) * C;
    v2d q[C];
#pragma unroll
    for (int c = 0; c < C; ++c) q[c] = mine[c];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int fr = (i <= j) ? i * N + j : j * N + i;
            const int fi = N * N + fr;
            z.re[i][j] = (fr & 1) ? q[fr >> 1].y : q[fr >> 1].x;
            z.im[i][j] = (fi & 1) ? q[fi >> 1].y : q[fi >> 1].x;
        }
    }
}

// Default for shallow grids (measured -1.9 % against waiting for both endpoints, tools/ab_bench.py; a
// 64-thread block instead of 256 was worth another 1 % and was not adopted): wait for the first endpoint only (vmcnt counts the 16 younger DMAs of the second endpoint),
// read it, let the caller's arithmetic on it start while the second endpoint is still landing.
template <int N>
__device__ __forceinline__ void gather_pair_dma_split(const double* __restrict__ base1, const int row1,
                                                      const double* __restrict__ base2, const int row2,
                                                      v2d* __restrict__ tile, sympa::CMat<N>& z1, sympa::CMat<N>& z2) {
    v2d* side1 = tile;
    v2d* side2 = tile + DmaTile<N>::SIDE_SLOTS;
    dma_issue<N>(base1, row1, side1);
    dma_issue<N>(base2, row2, side2);
    static_assert(N * N <= 16, "vmcnt immediate");
    if (N == 4) __builtin_amdgcn_s_waitcnt(0x4F70);   // vmcnt(16): the 16 DMAs of endpoint 1 have landed
    else __builtin_amdgcn_s_waitcnt(0x0070);
    wave_lds_fence();
    dma_read<N>(side1, z1);
    __builtin_amdgcn_s_waitcnt(0x0070);
    wave_lds_fence();
    dma_read<N>(side2, z2);
}

// (A packed shadow table -- 20 doubles per row, upper triangles only, 37 % fewer bytes -- was measured SLOWER,
// 7.26 vs 7.04 us per launch (tools/ab_packed.py at commit "packed experiment"): the gather is bound by the
// number of DMA instructions and the L2 round trip, not by bytes, and partially masked DMA instructions cost
// more than full ones.  Rows are fetched whole from the reference layout.)

// Low-LDS form: one endpoint at a time through one side buffer.  Exposes one more L2 round trip per
// wave but halves the LDS footprint, so twice as many waves fit on a CU (used when launches overlap or
// the grid is several waves per SIMD deep: the other wave's arithmetic hides the round trip).
template <int N>
__device__ __forceinline__ void gather_pair_dma_low(const double* __restrict__ base1, const int row1,
                                                    const double* __restrict__ base2, const int row2,
                                                    v2d* __restrict__ tile, sympa::CMat<N>& z1, sympa::CMat<N>& z2) {
    dma_issue<N>(base1, row1, tile);
    __builtin_amdgcn_s_waitcnt(0x0070);
    wave_lds_fence();
    dma_read<N>(tile, z1);
    wave_lds_fence();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): our reads of the buffer are done before it is refilled
    dma_issue<N>(base2, row2, tile);
    __builtin_amdgcn_s_waitcnt(0x0070);
    wave_lds_fence();
    dma_read<N>(tile, z2);
}

// Minimum-LDS form (n = 4): the 64 rows of an endpoint are gathered in 4 passes of 16 pairs -- pass p takes the
// pairs whose lane%16 is in [4p, 4p+4), so the row index still comes from the lane's own DPP row -- through two
// ping-pong buffers of 4 DMA instructions (4 160 B) each: 8.3 KB of LDS per wave instead of 16.6 / 33 KB, so that
// three blocks of different launches fit on a CU (the register budget, 140 VGPRs, allows three waves per SIMD; forcing 128 VGPRs for four waves cost 36 B of
// scratch and was slower: 11.13 vs 11.63 G pairs/s).  Measured with steps overlapped on 4 streams: 11.63 G pairs/s
// against 10.84 for the one-endpoint-at-a-time form on 2 streams.
template <int P, int J>
struct DmaIssuePass4 {
    static __device__ __forceinline__ void run(const double* __restrict__ base, const int row, const int c,
                                               const unsigned buf) {
        const int rr = group_bcast<16, 4 * P + J>(row);
        const unsigned off = (unsigned)rr * 256u + (unsigned)(16 * c);
        const char* src = reinterpret_cast<const char*>(base) + off;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, lds_dest_at(buf + (unsigned)(J * DmaTile<4>::INSTR_SLOTS) * 16u), 16, 0, 0);
        if constexpr (J + 1 < 4) DmaIssuePass4<P, J + 1>::run(base, row, c, buf);
    }
};

template <int P>
__device__ __forceinline__ void pass4_read(const v2d* __restrict__ buf, sympa::CMat<4>& z) {
    const int lane = threadIdx.x & 63;
    const int j = (lane & 15) - 4 * P;
    if (j >= 0 && j < 4) {
        const v2d* mine = buf + j * DmaTile<4>::INSTR_SLOTS + (lane >> 4) * 16;
        v2d q[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) q[c] = mine[c];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int fr = (i <= k) ? i * 4 + k : k * 4 + i;
                const int fi = 16 + fr;
                z.re[i][k] = (fr & 1) ? q[fr >> 1].y : q[fr >> 1].x;
                z.im[i][k] = (fi & 1) ? q[fi >> 1].y : q[fi >> 1].x;
            }
    }
}

constexpr int PASS4_BUF_SLOTS = 4 * 65;
constexpr int PASS4_WAVE_SLOTS = 2 * PASS4_BUF_SLOTS;

template <int S>
__device__ __forceinline__ void pass4_step(const double* __restrict__ base1, const int row1,
                                           const double* __restrict__ base2, const int row2, v2d* __restrict__ buf0,
                                           v2d* __restrict__ buf1, const int c, sympa::CMat<4>& z1, sympa::CMat<4>& z2) {
    v2d* cur = (S & 1) ? buf1 : buf0;
    v2d* nxt = (S & 1) ? buf0 : buf1;
    if constexpr (S + 1 < 8) {
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): reads of the buffer being refilled are complete
        wave_lds_fence();
        if constexpr (S + 1 < 4) DmaIssuePass4<S + 1, 0>::run(base1, row1, c, lds_offset_uniform(nxt));
        else DmaIssuePass4<S + 1 - 4, 0>::run(base2, row2, c, lds_offset_uniform(nxt));
        __builtin_amdgcn_s_waitcnt(0x0F74);   // vmcnt(4): the 4 DMAs of step S have landed
    } else {
        __builtin_amdgcn_s_waitcnt(0x0070);
    }
    wave_lds_fence();
    if constexpr (S < 4) pass4_read<S>(cur, z1);
    else pass4_read<S - 4>(cur, z2);
    if constexpr (S + 1 < 8) pass4_step<S + 1>(base1, row1, base2, row2, buf0, buf1, c, z1, z2);
}

__device__ __forceinline__ void gather_pair_pass4(const double* __restrict__ base1, const int row1,
                                                  const double* __restrict__ base2, const int row2,
                                                  v2d* __restrict__ tile, sympa::CMat<4>& z1, sympa::CMat<4>& z2) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { z1.re[i][j] = 0.0; z1.im[i][j] = 0.0; z2.re[i][j] = 0.0; z2.im[i][j] = 0.0; }
    const int c = (threadIdx.x & 63) & 15;
    DmaIssuePass4<0, 0>::run(base1, row1, c, lds_offset_uniform(tile));
    pass4_step<0>(base1, row1, base2, row2, tile, tile + PASS4_BUF_SLOTS, c, z1, z2);
}

// Both endpoints: all global loads are in flight before the first transpose starts.
template <int N>
__device__ __forceinline__ void gather_pair_staged(const double* __restrict__ base1, const int row1,
                                                   const double* __restrict__ base2, const int row2,
                                                   v2d* __restrict__ tile, sympa::CMat<N>& z1, sympa::CMat<N>& z2) {
    v2d v1[N * N], v2[N * N];
    gather_issue<N>(base1, row1, v1);
    gather_issue<N>(base2, row2, v2);
    gather_transpose<N>(v1, tile, z1);
    gather_transpose<N>(v2, tile, z2);
}

// ---------------------------------------------------------------------------------------------
// n = 5..8: rows are 400 B .. 1 KiB, a whole wave's tile would not fit in LDS.  The 64 rows of an endpoint are gathered in 4
// passes of 16 rows through a ring of LDS buffers: one LDS-DMA instruction fetches one row, later passes are in flight while
// the 16 lanes of pass p read their rows back (odd row pitch, conflict-free).  Measured need: with lane-per-row loads the
// n = 8 kernel was bound by L1/TA line thrashing (257 us per 262 144 pairs against ~75 us of arithmetic).
//
// Round 3 (profiles/r03_n8_forward_ab.txt): with one wave per SIMD nothing but the ring hides the latency of these loads --
// the upper n = 8 forward takes 125 us per 262 144 pairs on a table that fits the L2s and 154 us on configs[3]'s 46.6 MB
// table.  MASKED form: only the upper triangles are read afterwards, so a lane fetches only a 16-byte chunk that CONTAINS an
// upper-triangle element (40 of the 64 chunks of an 8 x 8 row; lane k fetches the k-th needed chunk, the DMA packs the
// active lanes' chunks back to back): a row is 640 B in LDS instead of 1 KiB, THREE buffers fit where two did, two passes
// are in flight behind the one being read, 37 % fewer bytes leave memory.  Measured A/B against the two-buffer full-row
// form: bounded n = 8 248 -> 238 us, upper n = 8 152.8 -> 157.5 us (and 123 -> 130 us on an L2-resident table: the select
// chain, the exec-masked DMA and the different register allocation cost more than the deeper ring hides), dims 5..7 equal.
// So the latency that shows is not the ring's steady state but the head of every wave (index load, then the first pass,
// nothing to overlap with) -- MASKED is used for the bounded model, the upper model keeps full rows and two buffers.
// ---------------------------------------------------------------------------------------------
template <int N, bool MASKED>
struct PassChunks {
    static constexpr int C = N * N;                       // 16-byte chunks per row ([2, N, N] doubles)
    struct Map {
        int count;
        int src[C];          // k-th needed chunk
        int slot[C];         // packed position of chunk c, -1 when not fetched
    };
    static constexpr Map make() {
        Map m{};
        bool need[C] = {};
        if (MASKED) {
            for (int pl = 0; pl < 2; ++pl)
                for (int i = 0; i < N; ++i)
                    for (int j = i; j < N; ++j) need[(pl * N * N + i * N + j) / 2] = true;
        } else {
            for (int c = 0; c < C; ++c) need[c] = true;
        }
        int k = 0;
        for (int c = 0; c < C; ++c) {
            m.slot[c] = -1;
            if (need[c]) { m.src[k] = c; m.slot[c] = k; ++k; }
        }
        for (int c = k; c < C; ++c) m.src[c] = 0;
        m.count = k;
        return m;
    }
    static constexpr Map MAP = make();
    static constexpr int K = MAP.count;
};

template <int N, bool MASKED = false>
struct PassTile {
    static constexpr int C = N * N;
    static constexpr bool ENABLED = (N >= 5 && N <= 8);
    static constexpr int K = PassChunks<N, MASKED>::K;                   // chunks of a row that are fetched
    static constexpr int PITCH = (K % 2 == 1) ? K : K + 1;
    static constexpr int ROWS = 16;
    static constexpr int BUF_SLOTS = ENABLED ? ROWS * PITCH : 1;
    static constexpr int NBUF = MASKED ? 3 : 2;
    static constexpr int WAVE_SLOTS = NBUF * BUF_SLOTS;
};

// the chunk lane `lane` fetches (a select chain over the constexpr map: once per wave)
template <int N, bool MASKED>
__device__ __forceinline__ int pass_my_chunk(const int lane) {
    if constexpr (!MASKED) return lane < N * N ? lane : 0;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < PassChunks<N, MASKED>::K; ++k) mine = (lane == k) ? PassChunks<N, MASKED>::MAP.src[k] : mine;
    return mine;
}

template <int N, bool MASKED>
__device__ __forceinline__ void pass_issue(const double* __restrict__ base, const int row, const int pass,
                                           v2d* __restrict__ buf, const int my_chunk) {
    constexpr int K = PassTile<N, MASKED>::K;
    const int lane = threadIdx.x & 63;
    // One lane mask around the pass instead of a save / restore of EXEC per instruction, the LDS destinations without a null check
    // (lds_dest); the row indices are read (v_readlane: a scalar base address per row) IN FRONT of the mask -- behind the branch the
    // compiler may compute `row` for the active lanes only, and a pass reads lanes K..63 as well.  Both models since the end of
    // round 5: the masked gather of the bounded model used __shfl per instruction (a ds_bpermute + wait + v_readfirstlane each)
    // because v_readlane alone had regressed its dims-8 kernel (245 -> 283 us); in this structure it gains -- bounded dense forward
    // per 262 144 pairs: n = 7 201.7 -> 139.7 us, n = 6 101.1 -> 95.2, n = 5 64.3 -> 56.5, n = 8 237.9 -> 237.7; upper: block 9 of
    // profiles/r05_packed_forward.txt.
    int rr[PassTile<N>::ROWS];
#pragma unroll
    for (int j = 0; j < PassTile<N>::ROWS; ++j) rr[j] = __builtin_amdgcn_readlane(row, 16 * pass + j);
    if (K == 64 || lane < K) {
#pragma unroll
        for (int j = 0; j < PassTile<N>::ROWS; ++j) {
            const double* src = base + (int64_t)rr[j] * (2 * N * N) + 2 * my_chunk;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, lds_dest(buf + j * PassTile<N, MASKED>::PITCH), 16, 0, 0);
        }
    }
}

template <int N, bool MASKED>
__device__ __forceinline__ void pass_read(const v2d* __restrict__ buf, const int pass, sympa::CMat<N>& z) {
    constexpr int K = PassTile<N, MASKED>::K;
    const int lane = threadIdx.x & 63;
    if ((lane >> 4) == pass) {
        const v2d* mine = buf + (lane & 15) * PassTile<N, MASKED>::PITCH;
        v2d q[K];
#pragma unroll
        for (int c = 0; c < K; ++c) q[c] = mine[c];
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int fr = (i <= j) ? i * N + j : j * N + i;
                const int fi = N * N + fr;
                z.re[i][j] = (fr & 1) ? q[PassChunks<N, MASKED>::MAP.slot[fr >> 1]].y : q[PassChunks<N, MASKED>::MAP.slot[fr >> 1]].x;
                z.im[i][j] = (fi & 1) ? q[PassChunks<N, MASKED>::MAP.slot[fi >> 1]].y : q[PassChunks<N, MASKED>::MAP.slot[fi >> 1]].x;
            }
        }
    }
}

// s_waitcnt vmcnt(V) only (lgkmcnt / expcnt untouched): V in [0, 63]
template <int V>
__device__ __forceinline__ void wait_vmcnt() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (V & 15) | ((V >> 4) << 14));
}

template <int N, bool MASKED>
__device__ __forceinline__ void gather_pair_passes(const double* __restrict__ base1, const int row1,
                                                   const double* __restrict__ base2, const int row2,
                                                   v2d* __restrict__ tile, sympa::CMat<N>& z1, sympa::CMat<N>& z2) {
    constexpr int NBUF = PassTile<N, MASKED>::NBUF;
    constexpr int LOOK = NBUF - 1;                  // passes in flight behind the one being read
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) { z1.re[i][j] = 0.0; z1.im[i][j] = 0.0; z2.re[i][j] = 0.0; z2.im[i][j] = 0.0; }
    const int my_chunk = pass_my_chunk<N, MASKED>(threadIdx.x & 63);
    auto issue = [&](const int p) {
        v2d* buf = tile + (p % NBUF) * PassTile<N, MASKED>::BUF_SLOTS;
        if (p < 4) pass_issue<N, MASKED>(base1, row1, p, buf, my_chunk);
        else pass_issue<N, MASKED>(base2, row2, p - 4, buf, my_chunk);
    };
#pragma unroll
    for (int p = 0; p < NBUF; ++p) issue(p);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // passes s+1 .. s+LOOK (those that exist) stay in flight; pass s has landed
        const int behind = (8 - 1 - s) < LOOK ? (8 - 1 - s) : LOOK;
        if (behind == 2) wait_vmcnt<32>();
        else if (behind == 1) wait_vmcnt<16>();
        else wait_vmcnt<0>();
        wave_lds_fence();
        const v2d* cur = tile + (s % NBUF) * PassTile<N, MASKED>::BUF_SLOTS;
        if (s < 4) pass_read<N, MASKED>(cur, s, z1);
        else pass_read<N, MASKED>(cur, s - 4, z2);
        if (s + NBUF < 8) {
            // the buffer just read is refilled: make sure those reads have completed
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
            wave_lds_fence();
            issue(s + NBUF);
        }
    }
}
