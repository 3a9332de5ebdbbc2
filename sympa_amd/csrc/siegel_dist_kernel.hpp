// Kernel templates of the one-pair-per-lane forward (dims 1..8) and their launchers, shared by two translation units:
// siegel_dist.hip instantiates dims 1..4, siegel_dist_big.hip dims 5..8 -- the latter is compiled without the pre-RA machine
// scheduler (-mllvm -enable-misched=0, __graft_entry__.py): the fully unrolled dims-8 kernel then keeps the source order and
// needs 400 fewer v_mov_b64 / 170 fewer v_accvgpr_read (profiles/r03_n8_forward_ab.txt), while the dims <= 4 kernels and the
// fused multi-batch kernel of the headline measured 0-6 % slower without it.
#pragma once
#include "siegel_common.hpp"
#include <hip/hip_ext.h>

namespace sympa_hip {

// Body of one 256-thread block: pairs [first, first + 256) of the batch described by `a`.
// dims 5..8: the bounded model gathers only the chunks that hold upper-triangle elements through three ring buffers, the
// upper model whole rows through two (siegel_gather.hpp: measured per model)
template <int MODEL>
constexpr bool pass_masked() { return MODEL != sympa::MODEL_UPPER; }

template <int N, int MODEL, bool LOWLDS>
struct BlockLds {
    static constexpr bool PASS4 = (N == 4) && LOWLDS;   // minimum-LDS gather: three blocks of different launches per CU
    static constexpr int WAVE_SLOTS = PASS4 ? PASS4_WAVE_SLOTS
                                      : DmaTile<N>::ENABLED ? (LOWLDS ? DmaTile<N>::WAVE_SLOTS_LOW : DmaTile<N>::WAVE_SLOTS)
                                      : (PassTile<N>::ENABLED ? PassTile<N, pass_masked<MODEL>()>::WAVE_SLOTS : Tile<N>::WAVE_SLOTS);
};

template <int N, int MODEL, bool LOWLDS>
__device__ __forceinline__ void dist_block(const DistArgs& a, const int64_t first, v2d* __restrict__ lds) {
    constexpr bool PASS4 = BlockLds<N, MODEL, LOWLDS>::PASS4;
    constexpr int WAVE_SLOTS = BlockLds<N, MODEL, LOWLDS>::WAVE_SLOTS;
    const int64_t i = first + threadIdx.x;
    const bool live = i < a.b;
    const int64_t ii = live ? i : a.b - 1;   // idle tail lanes recompute the last pair (wave ballots need them)

    int st = 0;
    int64_t r1 = ii, r2 = ii;
    if (a.ap_cols > 0) {          // all-pairs block of the distance matrix (runner.py:142-154)
        ap_pair(a, ii, r1, r2);
    } else if (a.idx1 != nullptr) {
        // index batches and outputs stream through once: non-temporal, so they do not evict table rows from L2
        r1 = __builtin_nontemporal_load(a.idx1 + ii * a.idx1_stride);
        r2 = __builtin_nontemporal_load(a.idx2 + ii * a.idx2_stride);
        if (r1 < 0 || r1 >= a.num_rows || r2 < 0 || r2 >= a.num_rows) {
            st |= sympa::ST_BAD_INDEX;
            r1 = 0;
            r2 = 0;
        }
    }
    constexpr int64_t ROW = 2 * N * N;
    double* vv = (a.vvd != nullptr && live) ? a.vvd + i * N : nullptr;
    double d;
    if constexpr (PassTile<N>::ENABLED) {
        sympa::CMat<N> z1, z2;
        v2d* tile = lds + (threadIdx.x >> 6) * WAVE_SLOTS;
        gather_pair_passes<N, pass_masked<MODEL>()>(a.base1, (int)r1, a.base2, (int)r2, tile, z1, z2);
        d = sympa::pair_distance_mats<N, MODEL>(z1, z2, a.metric, a.metric_w, a.inv_eps, vv, st);
    } else if constexpr (Tile<N>::STAGED) {
        sympa::CMat<N> z1, z2;
        v2d* tile = lds + (threadIdx.x >> 6) * WAVE_SLOTS;
        if constexpr (PASS4)
            gather_pair_pass4(a.base1, (int)r1, a.base2, (int)r2, tile, z1, z2);
        else if constexpr (DmaTile<N>::ENABLED && LOWLDS)
            gather_pair_dma_low<N>(a.base1, (int)r1, a.base2, (int)r2, tile, z1, z2);
        else if constexpr (DmaTile<N>::ENABLED)
            gather_pair_dma_split<N>(a.base1, (int)r1, a.base2, (int)r2, tile, z1, z2);
        else
            gather_pair_staged<N>(a.base1, (int)r1, a.base2, (int)r2, tile, z1, z2);
        d = sympa::pair_distance_mats<N, MODEL>(z1, z2, a.metric, a.metric_w, a.inv_eps, vv, st);
    } else {
        d = sympa::pair_distance<N, MODEL>(a.base1 + r1 * ROW, a.base2 + r2 * ROW, a.metric, a.metric_w,
                                           a.inv_eps, vv, st);
    }
    if (st & sympa::ST_BAD_INDEX) d = __builtin_nan("");
    if (a.scale != nullptr) d *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);   // model.py:40-41
    if (live) {
        if (a.ap_cols > 0) ap_store(a, i, r1, r2, d);
        else __builtin_nontemporal_store(d, a.out + i);
    }

    if (a.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {   // rare path
            if (flagged) atomicOr(&a.status[0], st);
            if ((threadIdx.x & 63) == 0) atomicAdd(&a.status[1], (int)__popcll(m));
        }
    }
}

// Waves per SIMD the compiler must make room for: 1 = one 512-register wave (dims 5, 7, 8).  dims 6, upper: 286 registers by
// itself; held to 256 (two waves per SIMD, 35 registers in scratch) it measures 88.4 -> 83.7 us per 262 144 pairs on the 46.6 MB
// table and 84.4 -> 74.4 us on an L2-resident one (profiles/r03_n6_two_waves.txt).  bounded would spill 163 registers, dims 7 and 8
// several hundred.
template <int N, int MODEL> constexpr int fwd_min_blocks() { return (N == 6 && MODEL == sympa::MODEL_UPPER) ? 2 : 1; }

template <int N, int MODEL, bool LOWLDS>
__global__ __launch_bounds__(fwd_block(N), (fwd_min_blocks<N, MODEL>())) void siegel_dist_kernel(const DistArgs a) {
    __shared__ v2d lds[(fwd_block(N) / 64) * BlockLds<N, MODEL, LOWLDS>::WAVE_SLOTS];
    // Staggered first round (dims 7, 8 on tables that do not fit the L2s; the launcher sets the bit): a lone launch starts one wave
    // on every SIMD at the same instant, all of them gather their 128 KB of rows at once, compute in step and come back for the
    // next round together -- 134 MB bursts on a fabric that then idles.  CU j of every XCD (blocks 32 j .. 32 j + 31 of the first
    // 1 024) starts j x 0.6 us late; the rounds stay out of step from there on.  Upper n = 8, 262 144 pairs, 45 500 rows:
    // 151.1 -> 134.4 us (profiles/r04_fwd_stagger.txt); flat 133.5-134 us for every table from 16 MB up, equal at 10 MB,
    // +4-8 % on L2-resident tables (hence the size test), no gain for the bounded model (more arithmetic per byte).
    if constexpr (N >= 7) {
        if ((a.flags & SYMPA_INTERNAL_FLAG_STAGGER) && blockIdx.x < 1024u) {
            const int k = (int)((blockIdx.x >> 5) & 31u);
            for (int j = 0; j < k; ++j) __builtin_amdgcn_s_sleep(20);
        }
    }
    dist_block<N, MODEL, LOWLDS>(a, (int64_t)blockIdx.x * fwd_block(N), lds);
}

// Several batches in ONE launch (C-ABI sympa_model_forward_batches with SYMPA_FLAG_FUSE): block x belongs to the
// batch k with blk_end[k-1] <= x < blk_end[k]; the batches share table, metric and scale and differ in their index
// list, size and output.  The grid is several blocks per CU deep, so the minimum-LDS gather form is used: three
// blocks of different batches share a CU and one batch's gather hides behind another's arithmetic -- what
// overlapping the launches on streams achieves, without kernel boundaries, fork/join dependencies or idle SIMDs
// when a batch is smaller than one wave per SIMD.
struct MultiArgs {
    DistArgs c;                                    // idx1/idx2/out/b are taken from the lists below
    const int64_t* trip[SYMPA_MAX_FUSED_BATCHES];
    double* out[SYMPA_MAX_FUSED_BATCHES];
    int64_t b[SYMPA_MAX_FUSED_BATCHES];
    unsigned blk_end[SYMPA_MAX_FUSED_BATCHES];     // exclusive prefix end, in blocks
    int num_batches;
};

// Waves per SIMD the compiler must make room for in the fused multi-batch kernel (the bounded n = 4 instantiation keeps its 146
// registers = three waves per SIMD: capped at 128 for four it spills 66 and runs 5.39 -> 8.03 us/step, profiles/r03_rejected_variants.txt)
template <int N, int MODEL>
constexpr int multi_min_blocks() { return (N == 6 && MODEL == sympa::MODEL_UPPER) ? 2 : 1; }

template <int N, int MODEL>
__global__ __launch_bounds__(fwd_block(N), (multi_min_blocks<N, MODEL>())) void siegel_dist_multi_kernel(const MultiArgs m) {
    constexpr bool LOW = DmaTile<N>::ENABLED;
    __shared__ v2d lds[(fwd_block(N) / 64) * BlockLds<N, MODEL, LOW>::WAVE_SLOTS];
    // batch of this block: binary search over <= 32 prefix ends (block-uniform, scalar)
    int lo = 0, hi = m.num_batches - 1;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int mid = (lo + hi) >> 1;
        const bool right = blockIdx.x >= m.blk_end[mid];
        lo = right ? mid + 1 : lo;
        hi = right ? hi : mid;
    }
    const int k = lo;
    const unsigned blk0 = (k == 0) ? 0u : m.blk_end[k - 1];
    DistArgs a = m.c;
    a.idx1 = m.trip[k];
    a.idx2 = m.trip[k] + 1;
    a.out = m.out[k];
    a.b = m.b[k];
    dist_block<N, MODEL, LOW>(a, (int64_t)(blockIdx.x - blk0) * fwd_block(N), lds);
}

// One launch.  SYMPA_FLAG_ANY_ORDER: the dispatch packet carries no barrier bit (hipExtAnyOrderLaunch), so the command
// processor may start it while earlier launches of the SAME stream are still running -- independent steps overlap
// without any cross-stream dependency.
template <typename K>
hipError_t launch_kernel(K kern, unsigned grid, int block, const DistArgs& a, hipStream_t s) {
    if (a.flags & SYMPA_FLAG_ANY_ORDER) {
        void* args[] = {const_cast<DistArgs*>(&a)};
        return hipExtLaunchKernel(reinterpret_cast<const void*>(kern), dim3(grid), dim3(block), args, 0, s, nullptr,
                                  nullptr, hipExtAnyOrderLaunch);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, s, a);
    return hipGetLastError();
}

template <int N>
int launch_n(const DistArgs& a, int model, hipStream_t s) {
    constexpr int FB = fwd_block(N);
    const unsigned grid = (unsigned)((a.b + FB - 1) / FB);
    DistArgs st = a;
    st.flags &= ~SYMPA_INTERNAL_FLAG_STAGGER;             // an internal bit: never taken from the caller
    if (N >= 7 && model == SYMPA_MODEL_UPPER && a.idx1 != nullptr && a.ap_cols == 0 && grid >= 2048u &&
        a.num_rows * (int64_t)(16 * N * N) >= ((int64_t)12 << 20) && !(a.flags & SYMPA_FLAG_ANY_ORDER))
        st.flags |= SYMPA_INTERNAL_FLAG_STAGGER;          // two rounds or more of a table beyond the L2s: see the kernel
    // low-LDS gather when asked for, or when the grid is deep enough for a second block per CU to matter
    const bool low = DmaTile<N>::ENABLED && ((a.flags & SYMPA_FLAG_LOW_LDS) || grid > 2 * 256 * (BLOCK / FB));
    hipError_t e;
    if (model == SYMPA_MODEL_UPPER) {
        if (low) e = launch_kernel(siegel_dist_kernel<N, sympa::MODEL_UPPER, true>, grid, FB, st, s);
        else e = launch_kernel(siegel_dist_kernel<N, sympa::MODEL_UPPER, false>, grid, FB, st, s);
    } else {
        if (low) e = launch_kernel(siegel_dist_kernel<N, sympa::MODEL_BOUNDED, true>, grid, FB, st, s);
        else e = launch_kernel(siegel_dist_kernel<N, sympa::MODEL_BOUNDED, false>, grid, FB, st, s);
    }
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

template <int N>
int launch_multi_n(const MultiArgs& m, unsigned grid, int model, hipStream_t s) {
    if (model == SYMPA_MODEL_UPPER)
        hipLaunchKernelGGL((siegel_dist_multi_kernel<N, sympa::MODEL_UPPER>), dim3(grid), dim3(fwd_block(N)), 0, s, m);
    else
        hipLaunchKernelGGL((siegel_dist_multi_kernel<N, sympa::MODEL_BOUNDED>), dim3(grid), dim3(fwd_block(N)), 0, s, m);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}


// siegel_dist_big.hip: dims 5..8
int launch_dist_big(const DistArgs& a, int n, int model, hipStream_t s);
int launch_multi_big(const MultiArgs& m, unsigned grid, int n, int model, hipStream_t s);

}  // namespace sympa_hip
