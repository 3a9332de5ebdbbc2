// Backward kernel template shared by the translation units siegel_bwd.hip (n <= 6) and
// siegel_bwd_n{7,8}_{upper,bounded}.hip.
#pragma once
#include "siegel_common.hpp"
#include "siegel_gather.hpp"
#include "siegel_math_bwd.hpp"

namespace sympa_hip {

struct BwdArgs {
    DistArgs f;              // forward arguments (f.out may be null)
    const double* go;        // [b] dLoss/d(out)
    double* g1;              // dense [b,2,n,n] or the grad table
    double* g2;              // dense [b,2,n,n] or the grad table
    double* gw;              // [n] accumulated, or null
    double* gscale;          // [1] accumulated, or null
    const double* graph_dist;   // fused loss: [b] graph distances (then `go` is ignored), or null
    double* loss;               // fused loss: [1] accumulated sum |(d/g)^2 - 1| * loss_scale
    double loss_scale;
    double* wave_partials;      // deterministic mode: [waves][2 + n] per-wave sums (loss, d loss / d scale, d loss / d w_k) are
                                // WRITTEN here instead of being added to loss / gscale / gw with atomics; a later kernel
                                // (sympa_segment_sum_rows) adds them up in a fixed order
    const int* chunk_flags;     // one-pair-per-lane kernels of dims 5..8 (siegel_bwd_list_kernel): one word per 64-pair chunk of the
                                // batch -- only the chunks whose word is non-zero are processed, by a small fixed grid that scans the
                                // words (the split backward hands its graded-spectrum waves over this way, siegel_bwd_split.hip); null: all
};

// (Scattering only the n(n+1) upper-triangle entries of the symmetric rows and mirroring afterwards was
// measured SLOWER, 43.7 vs 38.5 us per 65 536 pairs: the atomic wave-instructions lose their contiguous
// whole-row shape, which matters more than the 37 % fewer bytes.)
template <int N>
struct ScatterTile {
    static constexpr int ROWD = 2 * N * N;           // doubles per row
    static constexpr bool BY_PLANE = N >= 7;         // n >= 7: Re plane, then Im plane through half the tile
    static constexpr int CHUNK = BY_PLANE ? N * N : ROWD;
    static constexpr int PITCH = CHUNK + 1;          // odd pitch: conflict-free ds_write_b64 per lane
    static constexpr int WAVE_DOUBLES = 64 * PITCH;
};

// merge = true (SYMPA_FLAG_MERGE_SRC, whole-row tiles: n <= 6): runs of consecutive pairs with the same row are summed inside the
// tile first -- the sum stays in the row of the run's LAST pair, the other rows of the run become zero, and the loop below skips
// zeros: one atomic row per run instead of one per pair (batches sorted by source: ~13 pairs per source row at the headline shape).
template <int N>
__device__ __forceinline__ void scatter_add_rows(const sympa::CMat<N>& g, const int row, double* __restrict__ grad,
                                                 double* __restrict__ tile, const bool live, const bool merge = false) {
    constexpr int ROWD = ScatterTile<N>::ROWD, PITCH = ScatterTile<N>::PITCH;
    const int lane = threadIdx.x & 63;
    if constexpr (ScatterTile<N>::BY_PLANE) {
        constexpr int NN = N * N;
SYMPA_UNROLL
        for (int plane = 0; plane < 2; ++plane) {
            wave_lds_fence();
SYMPA_UNROLL
            for (int i = 0; i < N; ++i)
SYMPA_UNROLL
                for (int j = 0; j < N; ++j)
                    tile[lane * PITCH + i * N + j] = live ? (plane == 0 ? g.re[i][j] : g.im[i][j]) : 0.0;
            wave_lds_fence();
#pragma unroll 4
            for (int t = 0; t < NN; ++t) {
                const int gidx = t * 64 + lane;
                const int p = gidx / NN, e = gidx - p * NN;
                const double val = tile[p * PITCH + e];
                const int r = __shfl(row, p);
                if (val != 0.0) atomicAdd(grad + (int64_t)r * ROWD + plane * NN + e, val);
            }
        }
        return;
    }
    wave_lds_fence();
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            tile[lane * PITCH + i * N + j] = live ? g.re[i][j] : 0.0;
            tile[lane * PITCH + N * N + i * N + j] = live ? g.im[i][j] : 0.0;
        }
    wave_lds_fence();
    if constexpr (!ScatterTile<N>::BY_PLANE) {
        if (merge) {                                                  // wave-uniform
            // (the shuffle runs in ALL lanes, outside the ||: behind a short-circuit it executes under a divergent branch and a lane
            // whose neighbour is inactive reads 0)
            const int next_row = __shfl_down(row, 1);
            const bool ends = (lane == 63) | (next_row != row);
            const unsigned long long end_mask = __ballot(ends);
            for (int e0 = 0; e0 < ROWD; e0 += 64) {
                const int e = e0 + lane;
                const bool on = e < ROWD;
                double acc = 0.0;
#pragma unroll
                for (int p0 = 0; p0 < 64; p0 += 8) {
                    double v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = on ? tile[(p0 + j) * PITCH + e] : 0.0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        acc += v[j];
                        const bool end = (end_mask >> (p0 + j)) & 1ull;       // scalar
                        if (on) tile[(p0 + j) * PITCH + e] = end ? acc : 0.0;
                        acc = end ? 0.0 : acc;
                    }
                }
            }
            wave_lds_fence();
        }
    }
#pragma unroll 4
    for (int t = 0; t < ROWD; ++t) {
        const int gidx = t * 64 + lane;
        const int p = gidx / ROWD, e = gidx - p * ROWD;
        const double val = tile[p * PITCH + e];
        const int r = __shfl(row, p);
        if (val != 0.0) atomicAdd(grad + (int64_t)r * ROWD + e, val);
    }
}

// Per-pair gradient rows ([b, 2, n, n]: row i = pair i) written through the same tile: the 64 rows of a wave are contiguous
// in memory, so every store instruction covers 512 contiguous bytes instead of one 8-byte word in each of 64 rows.
template <int N>
__device__ __forceinline__ void store_rows_coalesced(const sympa::CMat<N>& g, double* __restrict__ out_wave, double* __restrict__ tile,
                                                     const bool zero, const int live_pairs) {
    constexpr int ROWD = ScatterTile<N>::ROWD, PITCH = ScatterTile<N>::PITCH;
    static_assert(!ScatterTile<N>::BY_PLANE, "whole rows through the tile: n <= 6");
    const int lane = threadIdx.x & 63;
    wave_lds_fence();
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            tile[lane * PITCH + i * N + j] = zero ? 0.0 : g.re[i][j];
            tile[lane * PITCH + N * N + i * N + j] = zero ? 0.0 : g.im[i][j];
        }
    wave_lds_fence();
    if (live_pairs == 64) {                          // wave-uniform: a full wave needs no predicate (a save / restore of EXEC and a
#pragma unroll 4                                       // branch per store otherwise: 256 scalar instructions per wave at n = 4)
        for (int t = 0; t < ROWD; ++t) {
            const int gidx = t * 64 + lane;
            const int p = gidx / ROWD, e = gidx - p * ROWD;
            __builtin_nontemporal_store(tile[p * PITCH + e], out_wave + gidx);
        }
        return;
    }
#pragma unroll 4
    for (int t = 0; t < ROWD; ++t) {
        const int gidx = t * 64 + lane;
        const int p = gidx / ROWD, e = gidx - p * ROWD;
        if (p < live_pairs) __builtin_nontemporal_store(tile[p * PITCH + e], out_wave + gidx);
    }
}

// The source-side rows of a wave with runs of equal source ids MERGED (SYMPA_FLAG_MERGE_SRC): the rows are staged as above, then
// lane e walks the wave's pairs in order with entry e of the running sum and stores it into the slot of the pair that ENDS a run
// (its id differs from the next pair's, or it is the wave's last live pair); the other slots of a run are not written.  Summation
// in pair order: bitwise reproducible.  `id`: the pair's RAW source id (an out-of-range id is a run of its own unless repeated:
// what the host's slot list computes from the same ids, ops.sorted_slots).
template <int N>
__device__ __forceinline__ void store_rows_merged(const sympa::CMat<N>& g, const int64_t id, double* __restrict__ out_wave,
                                                  double* __restrict__ tile, const bool zero, const int live_pairs) {
    constexpr int ROWD = ScatterTile<N>::ROWD, PITCH = ScatterTile<N>::PITCH;
    static_assert(!ScatterTile<N>::BY_PLANE, "whole rows through the tile: n <= 6");
    const int lane = threadIdx.x & 63;
    wave_lds_fence();
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) {
            tile[lane * PITCH + i * N + j] = zero ? 0.0 : g.re[i][j];
            tile[lane * PITCH + N * N + i * N + j] = zero ? 0.0 : g.im[i][j];
        }
    wave_lds_fence();
    const int lo = (int)(id & 0xffffffffll), hi = (int)(id >> 32);
    const int next_lo = __shfl_down(lo, 1), next_hi = __shfl_down(hi, 1);      // (both in all lanes: no short-circuit around a shuffle)
    const bool differs = (next_lo != lo) | (next_hi != hi);
    const bool ends = (lane < live_pairs) & ((lane + 1 >= live_pairs) | differs);
    const unsigned long long end_mask = __ballot(ends);                  // wave-uniform
    if constexpr (ROWD <= 32) {
        // both halves of the wave at once: lanes 0..31 walk pairs 0..31, lanes 32..63 pairs 32..63 (entry e = lane & 31).  The run
        // that crosses the middle: the lower half's unfinished sum is added IN FRONT of the upper half's first run, whose store waits
        // until the loop is over.  Reads in groups of eight ahead of the sums (the LDS latency of a lone wave, 64 times over, was
        // 2.7 us of a 21 us step at configs[0]).
        const int h = lane >> 5, e = lane & 31;
        const bool on = e < ROWD;
        const unsigned half_mask = (unsigned)(end_mask >> (32 * h));      // per lane: my half's run ends
        double acc = 0.0, first_sum = 0.0;
        int first_pos = -1;
#pragma unroll
        for (int q0 = 0; q0 < 32; q0 += 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = on ? tile[(32 * h + q0 + j) * PITCH + e] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + j;
                acc += v[j];
                if ((half_mask >> q) & 1u) {
                    if (h == 1 && first_pos < 0) {
                        first_sum = acc;
                        first_pos = 32 + q;
                    } else if (on) {
                        __builtin_nontemporal_store(acc, out_wave + (int64_t)(32 * h + q) * ROWD + e);
                    }
                    acc = 0.0;
                }
            }
        }
        const double carry = __shfl(acc, e);           // the lower half's unfinished run (0 when pair 31 ended one)
        if (h == 1 && first_pos >= 0 && on) __builtin_nontemporal_store(carry + first_sum, out_wave + (int64_t)first_pos * ROWD + e);
    } else {
        for (int e0 = 0; e0 < ROWD; e0 += 64) {
            const int e = e0 + lane;
            const bool on = e < ROWD;
            double acc = 0.0;
#pragma unroll 8
            for (int p = 0; p < live_pairs; ++p) {
                acc += on ? tile[p * PITCH + e] : 0.0;
                if ((end_mask >> p) & 1ull) {                                 // scalar condition
                    if (on) __builtin_nontemporal_store(acc, out_wave + (int64_t)p * ROWD + e);
                    acc = 0.0;
                }
            }
        }
    }
}

// n >= 7: the scatter tile of a wave is 25-33 KB (one plane at a time), so a block is one wave and four blocks share
// a CU; the adjoint's working
// set (E, H, its eigenvectors, the adjoints of all of them) does not fit the register file and spills to scratch.
template <int N>
constexpr int bwd_block() { return N >= 7 ? 64 : BLOCK; }

// n >= 7: the scatter is a separate (not inlined) device function.  Inlined into the 8 x 8 adjoint it made the backend
// take 7 minutes on the n = 7 kernel (418 s against 48 s for the same kernel without the scatter).
template <int N>
__device__ __attribute__((noinline)) void scatter_add_rows_outlined(const sympa::CMat<N>& g, const int row, double* __restrict__ grad,
                                                                    double* __restrict__ tile, const bool live) {
    scatter_add_rows<N>(g, row, grad, tile, live);
}

// (two 256-register waves per SIMD in the rows-out form at n <= 4: measured slower, 32.2 -> 34.7 us upper, 36.5 -> 51.0 us bounded --
// 60 spilled registers cost more than the second wave brings, profiles/r03_rejected_variants.txt)
// Round 6: upper model, n = 4 -- the two Cholesky factors (28 doubles) wait in the wave's LDS tile between the solves that form E
// and the back-substitutions (pair_backward's park / unpark): 294 -> <= 256 registers without scratch, two blocks per CU.
template <int N, int MODEL>
constexpr bool bwd_parks_factors() { return N == 4 && MODEL == sympa::MODEL_UPPER; }
template <int N, int MODEL, bool SCATTER>
constexpr int bwd_min_blocks() { return bwd_parks_factors<N, MODEL>() ? 2 : 1; }

template <int N, bool SCATTER>
struct BwdLds {
    static constexpr int GATHER_SLOTS = DmaTile<N>::ENABLED ? DmaTile<N>::WAVE_SLOTS_LOW : Tile<N>::WAVE_SLOTS;
    static constexpr bool ROWS_TILE = !SCATTER && !ScatterTile<N>::BY_PLANE;      // per-pair rows leave through the tile too (n <= 6)
    static constexpr int SCATTER_SLOTS = (SCATTER || ROWS_TILE) ? (ScatterTile<N>::WAVE_DOUBLES + 1) / 2 : 1;
    static constexpr int WAVE_SLOTS = GATHER_SLOTS > SCATTER_SLOTS ? GATHER_SLOTS : SCATTER_SLOTS;
};

// One wave's 64 pairs [i - lane, i - lane + 64) of the batch window: everything siegel_bwd_kernel does (the kernel below is this
// body once per wave; siegel_bwd_list_kernel runs it for the chunks of a list).  `tile`: the wave's LDS tile.
template <int N, int MODEL, bool SCATTER>
__device__ __forceinline__ void siegel_bwd_body(const BwdArgs& a, const DistArgs& f, const double* __restrict__ graph_dist,
                                                const int64_t i, v2d* __restrict__ tile) {
    constexpr bool ROWS_TILE = BwdLds<N, SCATTER>::ROWS_TILE;
    const bool live = i < f.b;
    const int64_t ii = live ? i : f.b - 1;

    int st = 0;
    int64_t r1 = ii, r2 = ii;
    int64_t raw1 = ii;                           // (the merged rows form compares the ids as given)
    if (f.idx1 != nullptr) {
        r1 = f.idx1[ii * f.idx1_stride];
        r2 = f.idx2[ii * f.idx2_stride];
        raw1 = r1;
        if (r1 < 0 || r1 >= f.num_rows || r2 < 0 || r2 >= f.num_rows) {
            st |= sympa::ST_BAD_INDEX;
            r1 = 0;
            r2 = 0;
        }
    }
    constexpr int64_t ROW = 2 * N * N;
    sympa::CMat<N> z1, z2;
    if constexpr (DmaTile<N>::ENABLED) {
        gather_pair_dma_low<N>(f.base1, (int)r1, f.base2, (int)r2, tile, z1, z2);
    } else if constexpr (Tile<N>::STAGED) {
        gather_pair_staged<N>(f.base1, (int)r1, f.base2, (int)r2, tile, z1, z2);
    } else {
        sympa::load_point<N>(f.base1 + r1 * ROW, z1);
        sympa::load_point<N>(f.base2 + r2 * ROW, z2);
    }
    // out = dist * sc,  sc = max(scale / coef, 0.1)   (model.py:40-41)
    double sc = 1.0;
    bool sc_active = false;
    if (f.scale != nullptr) {
        const double raw = f.scale[0] * f.inv_scale_coef;
        sc_active = raw > 0.1;
        sc = sc_active ? raw : 0.1;
    }
    sympa::CMat<N> g1, g2;
    double gw[N];
SYMPA_UNROLL
    for (int k = 0; k < N; ++k) gw[k] = 0.0;
    // every gradient is linear in go: run the adjoint with go = 1 and scale afterwards (the fused loss
    // needs the distance before it knows go)
    double dist;
    if constexpr (bwd_parks_factors<N, MODEL>()) {
        // the factors wait in the wave's tile ([entry][lane]: conflict-free) while the eigen stage and the products run
        constexpr int TRI = N * (N + 1) / 2;
        double* const ptile = reinterpret_cast<double*>(tile) + (threadIdx.x & 63);
        dist = sympa::pair_backward<N, MODEL>(
            z1, z2, f.metric, f.metric_w, f.inv_eps, 1.0, g1, g2, gw, st,
            [&](const int which, sympa::Tri<N, false>& l) {
                double* p = ptile + which * TRI * 64;
SYMPA_UNROLL
                for (int r = 0; r < N; ++r) {
                    p[sympa::tri_index(N, r, r) * 64] = l.rdiag[r];
SYMPA_UNROLL
                    for (int c = 0; c < r; ++c) p[sympa::tri_index(N, c, r) * 64] = l.re[r][c];
                }
            },
            [&](const int which, sympa::Tri<N, false>& l) {
                const double* p = ptile + which * TRI * 64;
SYMPA_UNROLL
                for (int r = 0; r < N; ++r) {
                    l.rdiag[r] = p[sympa::tri_index(N, r, r) * 64];
SYMPA_UNROLL
                    for (int c = 0; c < r; ++c) l.re[r][c] = p[sympa::tri_index(N, c, r) * 64];
                }
            });
    } else {
        dist = sympa::pair_backward<N, MODEL>(z1, z2, f.metric, f.metric_w, f.inv_eps, 1.0, g1, g2, gw, st);
    }
    const bool bad = (st & sympa::ST_BAD_INDEX) != 0;
    double go = 0.0, loss_i = 0.0;
    if (graph_dist != nullptr) {   // AverageDistortionLoss (losses.py:10-19): sum |(d/g)^2 - 1|
        const double gd = live ? graph_dist[i] : 1.0;
        const double ratio = dist * sc / gd;
        const double e = ratio * ratio - 1.0;
        loss_i = (live && !bad) ? fabs(e) * a.loss_scale : 0.0;
        go = (e > 0.0 ? 1.0 : (e < 0.0 ? -1.0 : 0.0)) * 2.0 * ratio / gd * a.loss_scale;
        if (!live) go = 0.0;
    } else {
        go = live ? a.go[i] : 0.0;
    }
    {
        const double gs_ = go * sc;
SYMPA_UNROLL
        for (int r = 0; r < N; ++r)
SYMPA_UNROLL
            for (int c = 0; c < N; ++c) {
                g1.re[r][c] *= gs_; g1.im[r][c] *= gs_;
                g2.re[r][c] *= gs_; g2.im[r][c] *= gs_;
            }
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) gw[k] *= gs_;
    }
    if (live && f.out != nullptr) f.out[i] = bad ? __builtin_nan("") : dist * sc;

    if constexpr (SCATTER) {
        double* dtile = reinterpret_cast<double*>(tile);
        if constexpr (N >= 7) {
            scatter_add_rows_outlined<N>(g1, (int)r1, a.g1, dtile, live && !bad);
            scatter_add_rows_outlined<N>(g2, (int)r2, a.g2, dtile, live && !bad);
        } else {
            scatter_add_rows<N>(g1, (int)r1, a.g1, dtile, live && !bad, (f.flags & SYMPA_FLAG_MERGE_SRC) != 0);
            scatter_add_rows<N>(g2, (int)r2, a.g2, dtile, live && !bad);
        }
    } else if constexpr (ROWS_TILE) {
        // per-pair rows (a pair with an out-of-range index contributes zeros, like the scatter form skips it)
        double* dtile = reinterpret_cast<double*>(tile);
        const int64_t wave_first = i - (threadIdx.x & 63);
        const int64_t left = f.b - wave_first;
        const int live_pairs = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
        if (f.flags & SYMPA_FLAG_MERGE_SRC) store_rows_merged<N>(g1, raw1, a.g1 + wave_first * ROW, dtile, bad, live_pairs);
        else store_rows_coalesced<N>(g1, a.g1 + wave_first * ROW, dtile, bad, live_pairs);
        store_rows_coalesced<N>(g2, a.g2 + wave_first * ROW, dtile, bad, live_pairs);
    } else if (live) {
SYMPA_UNROLL
        for (int r = 0; r < N; ++r)
SYMPA_UNROLL
            for (int c = 0; c < N; ++c) {
                a.g1[i * ROW + r * N + c] = bad ? 0.0 : g1.re[r][c];
                a.g1[i * ROW + N * N + r * N + c] = bad ? 0.0 : g1.im[r][c];
                a.g2[i * ROW + r * N + c] = bad ? 0.0 : g2.re[r][c];
                a.g2[i * ROW + N * N + r * N + c] = bad ? 0.0 : g2.im[r][c];
            }
    }
    // reductions over the wave: one atomic per wave, or (deterministic mode) the wave's sums stored for a fixed-order sum
    if (a.wave_partials != nullptr) {
        // a wave with no live pair (the tail of the last 256-thread block) has no row in the [ceil(b / 64)][2 + n] buffer
        const bool writer = (threadIdx.x & 63) == 0 && i < f.b;
        double* wp = a.wave_partials + (i >> 6) * (2 + N);
        double x = loss_i;
        double y = (live && !bad && sc_active) ? go * dist * f.inv_scale_coef : 0.0;
SYMPA_UNROLL
        for (int off = 32; off > 0; off >>= 1) { x += __shfl_xor(x, off); y += __shfl_xor(y, off); }
        if (writer) { wp[0] = x; wp[1] = y; }
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) {
            double w = (live && !bad && f.metric == sympa::METRIC_WSUM) ? gw[k] : 0.0;
SYMPA_UNROLL
            for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off);
            if (writer) wp[2 + k] = w;
        }
    } else {
    if (a.gw != nullptr && f.metric == sympa::METRIC_WSUM) {
SYMPA_UNROLL
        for (int k = 0; k < N; ++k) {
            double x = (live && !bad) ? gw[k] : 0.0;
SYMPA_UNROLL
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if ((threadIdx.x & 63) == 0 && x != 0.0) atomicAdd(a.gw + k, x);
        }
    }
    if (a.gscale != nullptr && f.scale != nullptr) {
        double x = (live && !bad && sc_active) ? go * dist * f.inv_scale_coef : 0.0;
SYMPA_UNROLL
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0 && x != 0.0) atomicAdd(a.gscale, x);
    }
    if (a.loss != nullptr && graph_dist != nullptr) {
        double x = loss_i;
SYMPA_UNROLL
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0 && x != 0.0) atomicAdd(a.loss, x);
    }
    }
    if (f.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {
            if (flagged) atomicOr(&f.status[0], st);
            if ((threadIdx.x & 63) == 0) atomicAdd(&f.status[1], (int)__popcll(m));
        }
    }
}

__device__ __forceinline__ void bwd_batch_window(const BwdArgs& a, DistArgs& f, const double*& graph_dist) {
    f = a.f;
    graph_dist = a.graph_dist;
    if (f.batch_counter != nullptr) {            // the training graph's batch window (see DistArgs)
        const int64_t off = f.batch_counter[0] * f.b;
        f.idx1 += off * f.idx1_stride;
        f.idx2 += off * f.idx2_stride;
        if (graph_dist != nullptr) graph_dist += off;
    }
}

template <int N, int MODEL, bool SCATTER>
__global__ __launch_bounds__(bwd_block<N>(), (bwd_min_blocks<N, MODEL, SCATTER>())) void siegel_bwd_kernel(const BwdArgs a) {
    constexpr int BLOCK = bwd_block<N>();
    constexpr int WAVE_SLOTS = BwdLds<N, SCATTER>::WAVE_SLOTS;
    __shared__ v2d lds[(BLOCK / 64) * WAVE_SLOTS];
    DistArgs f;
    const double* graph_dist;
    bwd_batch_window(a, f, graph_dist);
    siegel_bwd_body<N, MODEL, SCATTER>(a, f, graph_dist, (int64_t)blockIdx.x * BLOCK + threadIdx.x, lds + (threadIdx.x >> 6) * WAVE_SLOTS);
}

// The same body over the flagged chunks of a.chunk_flags (dims 5..8): a fixed small grid; wave w reads the words [64 (w + k W),
// 64 (w + k W) + 64), one per lane, and runs the body for every set one (wave-uniform loop over the ballot).  Launched by the split
// backward for the waves its first stage flagged (graded spectra): usually none at all, and the 64 waves of this grid cost a few
// microseconds where a full grid of the (spilling, scratch-heavy) dims 7, 8 kernels whose waves return at once cost ~30.  No counter,
// no atomics, nothing to reset between launches: every wave of stage 1 writes its own word.
constexpr int BWD_LIST_WAVES = 64;
template <int N, int MODEL, bool SCATTER>
__global__ __launch_bounds__(bwd_block<N>(), 1) void siegel_bwd_list_kernel(const BwdArgs a) {
    constexpr int BLOCK = bwd_block<N>();
    constexpr int WAVE_SLOTS = BwdLds<N, SCATTER>::WAVE_SLOTS;
    __shared__ v2d lds[(BLOCK / 64) * WAVE_SLOTS];
    DistArgs f;
    const double* graph_dist;
    bwd_batch_window(a, f, graph_dist);
    const int chunks = (int)((f.b + 63) / 64);
    const int waves = (int)gridDim.x * (BLOCK / 64);
    const int lane = (int)(threadIdx.x & 63);
    for (int base = 64 * ((int)blockIdx.x * (BLOCK / 64) + (int)(threadIdx.x >> 6)); base < chunks; base += 64 * waves) {
        const int word = (base + lane < chunks) ? a.chunk_flags[base + lane] : 0;
        unsigned long long todo = __ballot(word != 0);                  // wave-uniform
        while (todo != 0ull) {
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            wave_lds_fence();
            siegel_bwd_body<N, MODEL, SCATTER>(a, f, graph_dist, (int64_t)(base + k) * 64 + lane, lds + (threadIdx.x >> 6) * WAVE_SLOTS);
            __builtin_amdgcn_s_waitcnt(0);
            wave_lds_fence();
        }
    }
}

template <int N, int MODEL, bool SCATTER>
int launch_bwd_nms(const BwdArgs& a, hipStream_t s) {
    constexpr int BLOCK = bwd_block<N>();
    if constexpr (N >= 5) {
        if (a.chunk_flags != nullptr) {
            hipLaunchKernelGGL((siegel_bwd_list_kernel<N, MODEL, SCATTER>), dim3(BWD_LIST_WAVES / (BLOCK / 64)), dim3(BLOCK), 0, s, a);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
        }
    }
    const unsigned grid = (unsigned)((a.f.b + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL((siegel_bwd_kernel<N, MODEL, SCATTER>), dim3(grid), dim3(BLOCK), 0, s, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

template <int N, int MODEL>
int launch_bwd_nm(const BwdArgs& a, bool scatter, hipStream_t s) {
    return scatter ? launch_bwd_nms<N, MODEL, true>(a, s) : launch_bwd_nms<N, MODEL, false>(a, s);
}

template <int N>
int launch_bwd_n(const BwdArgs& a, int model, bool scatter, hipStream_t s) {
    return model == SYMPA_MODEL_UPPER ? launch_bwd_nm<N, sympa::MODEL_UPPER>(a, scatter, s)
                                      : launch_bwd_nm<N, sympa::MODEL_BOUNDED>(a, scatter, s);
}

// n = 7, 8: one translation unit per kernel -- siegel_bwd_n{7,8}_{upper,bounded}_{scatter,dense}.hip -- because the
// fully unrolled adjoint of an 8 x 8 pair takes a minute or two to compile; the build compiles the units in parallel.
#define SYMPA_BWD_LARGE(N, M) \
    int launch_bwd_n##N##_##M##_scatter(const BwdArgs& a, hipStream_t s); \
    int launch_bwd_n##N##_##M##_dense(const BwdArgs& a, hipStream_t s); \
    inline int launch_bwd_n##N##_##M(const BwdArgs& a, bool scatter, hipStream_t s) { \
        return scatter ? launch_bwd_n##N##_##M##_scatter(a, s) : launch_bwd_n##N##_##M##_dense(a, s); \
    }
SYMPA_BWD_LARGE(7, upper)
SYMPA_BWD_LARGE(7, bounded)
SYMPA_BWD_LARGE(8, upper)
SYMPA_BWD_LARGE(8, bounded)
#undef SYMPA_BWD_LARGE

// dims 5..8 with eight lanes per pair (siegel_bwd_half*.hip; A/B and, where faster, the default)
int launch_bwd_half(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s);

// dims 9..16: siegel_bwd_rolled.hip (the same adjoint with rolled loops over scratch arrays)
int launch_bwd_rolled(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s);

}  // namespace sympa_hip
