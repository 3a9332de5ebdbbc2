// The two kernels of the split Siegel backward, dims 5..8, one pair per lane (arithmetic and rationale: siegel_math_bwd_split.hpp).
//   siegel_bwd_spectral_kernel: gather, factors, E, H, eigen-decomposition with vectors, metric value, fused loss (or the
//        caller's grad_out), Hbar / K scaled by go * scale into the workspace [AdjPack::LEN][ws_stride]; forward values, loss,
//        scale / weight gradients, status exactly as siegel_bwd_kernel.
//   siegel_bwd_gradient_kernel: gather AGAIN, factors and E again, the pack, products / solves / congruences, then the
//        atomic scatter into the table gradient (or per-pair rows).
// Both run one-wave blocks at one 512-register wave per SIMD (like the dims 5..8 forward kernels).
#pragma once

#include "siegel_bwd_kernel.hpp"
#include "siegel_math_bwd_split.hpp"

namespace sympa_hip {

struct SplitArgs {
    BwdArgs a;
    double* ws;              // [AdjPack<N, MODEL>::LEN][ws_stride] fp64
    int64_t ws_stride;       // >= b
    int* graded;             // [wave]: stage 1 writes 1 for a wave of 64 pairs it hands to the one-stage kernel (graded spectrum), 0
                             // otherwise (BwdArgs::chunk_flags of the third launch); nullptr: no such hand-over
};

// entry base (wave-uniform: a scalar register pair) + 32-bit byte offset of the lane's pair: global_load/store ... v_off, s[base]
__device__ __forceinline__ double* ws_at(double* entry, const unsigned byte_off) {
    return reinterpret_cast<double*>(reinterpret_cast<char*>(entry) + byte_off);
}

// fp64 atomic add (no return) at a wave-uniform base + the lane's 32-bit byte offset, in the global address space (a pointer that
// went through an asm statement is generic to the compiler: it would emit flat_atomic_add_f64)
__device__ __forceinline__ void global_add_f64(double* base, const unsigned byte_off, const double v) {
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble* p = (gdouble*)((gchar*)base + byte_off);
    (void)__builtin_amdgcn_global_atomic_fadd_f64(p, v);
}

// The staging tile of a wave: one plane (n x n doubles) of each of its 64 pairs, pair p at tile[p * PITCH ...], PITCH odd: the
// per-lane writes of stage_plane and the per-instruction reads of the flushes are both conflict-free.
template <int N>
__device__ __forceinline__ void stage_plane(const double (&m)[N][N], double* __restrict__ tile, const bool live) {
    constexpr int PITCH = N * N + 1;
    const int lane = threadIdx.x & 63;
    wave_lds_fence();
SYMPA_UNROLL
    for (int i = 0; i < N; ++i)
SYMPA_UNROLL
        for (int j = 0; j < N; ++j) tile[lane * PITCH + i * N + j] = live ? m[i < j ? i : j][i < j ? j : i] : 0.0;
    wave_lds_fence();
}

// sign * (the staged plane) added to the table gradient: instruction t adds the doubles [64 t, 64 t + 64) of the wave's 64 planes
// laid end to end, so consecutive lanes add to consecutive doubles (the cost of an fp64 atomic instruction goes with the 128-byte
// lines it touches).  `grad` points at the plane inside row 0.  The LDS reads are issued CHUNK at a time ahead of their atomics
// (one wave per SIMD: nothing else hides an LDS round trip), no per-value test (a dead or out-of-range pair adds zeros to a valid
// row), and at n = 8 a plane is exactly one instruction: its row is a scalar (v_readlane), the address a scalar base + lane offset.
template <int N>
__device__ __forceinline__ void flush_plane_atomic(const double sign, const int row, double* __restrict__ grad, const double* __restrict__ tile) {
    constexpr int NN = N * N, ROWD = 2 * N * N, PITCH = NN + 1;
    const int lane = threadIdx.x & 63;
    constexpr int CHUNK = 8;
    if constexpr (NN == 64) {
        // Consecutive pairs with the SAME row are summed in registers and leave as ONE instruction: a batch sorted by its first
        // column (sympa_amd/train_step.py sorts every batch it loads; the order inside a batch is free) has ~6 pairs per source
        // row at configs[3], which takes the source side from 128 to ~25 atomic instructions per pair-row.  The memory side
        // retires these atomics at ~2.8 G instructions/s whatever their lane mask (tools/microbench/atomic_rate.hip: 369 us for the
        // 1 M plane-instructions of 262 144 pairs -- more than this kernel's arithmetic), so every instruction saved counts.
        // The rows are scalars (v_readlane) and the comparison a scalar branch; dead pairs repeat the last live row with zeros.
        int prev = __builtin_amdgcn_readlane(row, 0);
        double acc = 0.0;
SYMPA_UNROLL
        for (int t0 = 0; t0 < 64; t0 += CHUNK) {
            double val[CHUNK];
SYMPA_UNROLL
            for (int u = 0; u < CHUNK; ++u) val[u] = tile[(t0 + u) * PITCH + lane];
SYMPA_UNROLL
            for (int u = 0; u < CHUNK; ++u) {
                const int r = __builtin_amdgcn_readlane(row, t0 + u);
                if (r != prev) {                                  // wave-uniform
                    double* base = grad + (int64_t)prev * ROWD;
                    // opaque scalar: otherwise the compiler folds the lane offset into a vector base, keeps all 64 row addresses
                    // for the next plane of the same rows and spills them
                    asm volatile("" : "+s"(base));
                    global_add_f64(base, (unsigned)lane * 8u, sign * acc);      // global_atomic_add_f64 v_off, v_data, s[base]
                    acc = 0.0;
                }
                prev = r;
                acc += val[u];
            }
        }
        {
            double* base = grad + (int64_t)prev * ROWD;
            asm volatile("" : "+s"(base));
            global_add_f64(base, (unsigned)lane * 8u, sign * acc);
        }
    } else {
        constexpr int TOTAL = NN;                           // instructions: 64 planes x NN doubles / 64 lanes
#pragma unroll 1
        for (int t0 = 0; t0 < TOTAL; t0 += CHUNK) {
            double val[CHUNK];
            int rr[CHUNK], ee[CHUNK];
SYMPA_UNROLL
            for (int u = 0; u < CHUNK; ++u) {
                const int t = (t0 + u < TOTAL) ? t0 + u : TOTAL - 1;
                const int gidx = t * 64 + lane;
                const int p = gidx / NN;
                ee[u] = gidx - p * NN;
                val[u] = tile[p * PITCH + ee[u]];
                rr[u] = __shfl(row, p);
            }
SYMPA_UNROLL
            for (int u = 0; u < CHUNK; ++u)
                if (t0 + u < TOTAL) unsafeAtomicAdd(grad + (int64_t)rr[u] * ROWD + ee[u], sign * val[u]);
        }
    }
}

// sign * (the staged plane) written to the per-pair gradient rows of the wave ([b, 2, n, n]: `out` points at the plane inside the
// wave's first row, live_pairs of its 64 rows exist): every store instruction covers 64 consecutive doubles of the planes laid end
// to end -- whole 128-byte lines instead of one 8-byte word in each of 64 rows
template <int N>
__device__ __forceinline__ void flush_plane_rows(const double sign, double* __restrict__ out, const double* __restrict__ tile, const int live_pairs) {
    constexpr int NN = N * N, ROWD = 2 * N * N, PITCH = NN + 1;
    const int lane = threadIdx.x & 63;
#pragma unroll 8
    for (int t = 0; t < NN; ++t) {
        const int gidx = t * 64 + lane;
        const int p = gidx / NN, e = gidx - p * NN;
        if (p < live_pairs) __builtin_nontemporal_store(sign * tile[p * PITCH + e], out + (int64_t)p * ROWD + e);
    }
}

// Staggered first round (cf. siegel_dist_kernel.hpp): the first 1 024 blocks of a launch start one wave on every SIMD at the same
// instant, and waves that start together stay together -- all of them compute for the same ~60 us, then all of them issue their
// 256 fp64 atomic instructions at once (the gradient kernel) or their 100 workspace stores (the spectral kernel).  The atomic
// units take ~250 us for the 67 M adds of 262 144 pairs and idle while everybody computes: kernel time = compute + atomics.
// Spread over one compute period the two overlap.  Block x of the first 1 024 sleeps (x * 5 mod STEPS) x ~2.7 us.
constexpr int SPLIT_STAGGER_STEPS_GRAD = 24;
constexpr int SPLIT_STAGGER_STEPS_SPEC = -40;
template <int STEPS>
__device__ __forceinline__ void split_stagger(const bool on) {
    // STEPS < 0: the forward's scheme (siegel_dist_kernel.hpp) -- CU j of every XCD (blocks 32 j .. 32 j + 31 of the first 1 024) starts
    // j x s_sleep(-STEPS) late: what spreads a GATHER burst (the spectral kernel: 703 -> 689 us fused step at -40; -20: 698, -60: 708)
    if constexpr (STEPS < 0) {
        if (on && blockIdx.x < 1024u) {
            const int k = (int)((blockIdx.x >> 5) & 31u);
            for (int j = 0; j < k; ++j) __builtin_amdgcn_s_sleep(-STEPS);
        }
    }
    if constexpr (STEPS > 0) {
        if (on && blockIdx.x < 1024u) {
            const int k = (int)((blockIdx.x * 5u) % (unsigned)STEPS);
            for (int j = 0; j < k; ++j) __builtin_amdgcn_s_sleep(100);
        }
    }
}

template <int N>
__device__ __forceinline__ void split_rows(const DistArgs& f, const int64_t ii, int64_t& r1, int64_t& r2, int& st) {
    r1 = ii; r2 = ii;
    if (f.idx1 != nullptr) {
        r1 = f.idx1[ii * f.idx1_stride];
        r2 = f.idx2[ii * f.idx2_stride];
        if (r1 < 0 || r1 >= f.num_rows || r2 < 0 || r2 >= f.num_rows) {
            st |= sympa::ST_BAD_INDEX;
            r1 = 0;
            r2 = 0;
        }
    }
}

template <int N, int MODEL>
__global__ __launch_bounds__(64, 1) void siegel_bwd_spectral_kernel(const SplitArgs sa) {
    using P = sympa::AdjPack<N, MODEL>;
    constexpr int WAVE_SLOTS = PassTile<N, false>::WAVE_SLOTS;
    __shared__ v2d lds[WAVE_SLOTS];
    split_stagger<SPLIT_STAGGER_STEPS_SPEC>((sa.a.f.flags & SYMPA_INTERNAL_FLAG_STAGGER) != 0);      // set by launch_bwd_split
    const BwdArgs& a = sa.a;
    DistArgs f = a.f;
    const double* graph_dist = a.graph_dist;
    if (f.batch_counter != nullptr) {            // the training graph's batch window (see DistArgs)
        const int64_t off = f.batch_counter[0] * f.b;
        f.idx1 += off * f.idx1_stride;
        f.idx2 += off * f.idx2_stride;
        if (graph_dist != nullptr) graph_dist += off;
    }
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = i < f.b;
    const int64_t ii = live ? i : f.b - 1;
    int st = 0;
    int64_t r1, r2;
    split_rows<N>(f, ii, r1, r2, st);
    double pack[P::LEN];
    double gw[N];
SYMPA_UNROLL
    for (int k = 0; k < N; ++k) gw[k] = 0.0;
    double dist;
    bool redo = false;
    {
        sympa::CMat<N> z1, z2;
        gather_pair_passes<N, false>(f.base1, (int)r1, f.base2, (int)r2, lds, z1, z2);
        // (`refine`: a pair of this wave has a graded spectrum under fone / fmin / wsum -- siegel_math_bwd_split.hpp, "Graded
        // spectra".  The kernel only takes note: the wave's 64 pairs are handed to the one-stage kernel, see below.)
        dist = sympa::pair_adjoint_spectral<N, MODEL>(z1, z2, f.metric, f.metric_w, f.inv_eps, pack, gw, st,
                                                      [&](sympa::CMat<N>&, double (&)[N]) { redo = true; });
    }
    // Graded spectrum somewhere in this wave (wave-uniform: the branch that set `redo` was taken by all lanes or none): E is gone, so
    // the eigenvalues cannot be refined to Rayleigh quotients here.  The wave hands on ZERO packs (the gradient kernel then adds /
    // writes zeros for its pairs), raises its flag and contributes nothing else; the third launch of launch_bwd_split runs the
    // one-stage kernel (pair_backward: quotients ||E v_i||^2 for every eigenvalue) on exactly the flagged waves.
    if (threadIdx.x == 0 && sa.graded != nullptr) sa.graded[blockIdx.x] = redo ? 1 : 0;
    if (redo && sa.graded != nullptr) {
        if (live) {
            const unsigned wo = (unsigned)i * 8u;
SYMPA_UNROLL
            for (int k = 0; k < P::LEN; ++k) *ws_at(sa.ws + k * sa.ws_stride, wo) = 0.0;
        }
        return;
    }
    double sc = 1.0;
    bool sc_active = false;
    if (f.scale != nullptr) {
        const double raw = f.scale[0] * f.inv_scale_coef;
        sc_active = raw > 0.1;
        sc = sc_active ? raw : 0.1;
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
    const double gs_ = go * sc;
    if (live) {
        // a pair with an out-of-range index hands on zeros: the gradient stage then adds / writes nothing for it
        const unsigned wo = (unsigned)i * 8u;
SYMPA_UNROLL
        for (int k = 0; k < P::LEN; ++k) *ws_at(sa.ws + k * sa.ws_stride, wo) = bad ? 0.0 : pack[k] * gs_;
        if (f.out != nullptr) f.out[i] = bad ? __builtin_nan("") : dist * sc;
    }
SYMPA_UNROLL
    for (int k = 0; k < N; ++k) gw[k] *= gs_;

    // reductions over the wave: one atomic per wave, or (deterministic mode) the wave's sums stored for a fixed-order sum
    if (a.wave_partials != nullptr) {
        const bool writer = threadIdx.x == 0 && i < f.b;
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
                if (threadIdx.x == 0 && x != 0.0) atomicAdd(a.gw + k, x);
            }
        }
        if (a.gscale != nullptr && f.scale != nullptr) {
            double x = (live && !bad && sc_active) ? go * dist * f.inv_scale_coef : 0.0;
SYMPA_UNROLL
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (threadIdx.x == 0 && x != 0.0) atomicAdd(a.gscale, x);
        }
        if (a.loss != nullptr && graph_dist != nullptr) {
            double x = loss_i;
SYMPA_UNROLL
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (threadIdx.x == 0 && x != 0.0) atomicAdd(a.loss, x);
        }
    }
    if (f.status != nullptr) {
        const int flagged = (live && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {
            if (flagged) atomicOr(&f.status[0], st);
            if (threadIdx.x == 0) atomicAdd(&f.status[1], (int)__popcll(m));
        }
    }
}

template <int N, int MODEL, bool SCATTER>
__global__ __launch_bounds__(64, 1) void siegel_bwd_gradient_kernel(const SplitArgs sa) {
    using P = sympa::AdjPack<N, MODEL>;
    constexpr int GATHER_SLOTS = PassTile<N, false>::WAVE_SLOTS;
    constexpr int PLANE_SLOTS = (64 * (N * N + 1) + 1) / 2 > 64 * (N * (N + 1) / 2) ? (64 * (N * N + 1) + 1) / 2
                                                                                  : 64 * (N * (N + 1) / 2);   // scatter_add_plane tile | the two parked factors
    constexpr int SCATTER_SLOTS = MODEL == sympa::MODEL_UPPER ? PLANE_SLOTS : (SCATTER ? (ScatterTile<N>::WAVE_DOUBLES + 1) / 2 : 1);
    constexpr int WAVE_SLOTS = GATHER_SLOTS > SCATTER_SLOTS ? GATHER_SLOTS : SCATTER_SLOTS;
    __shared__ v2d lds[WAVE_SLOTS];
    split_stagger<SPLIT_STAGGER_STEPS_GRAD>(gridDim.x >= 2048u);
    const BwdArgs& a = sa.a;
    DistArgs f = a.f;
    if (f.batch_counter != nullptr) {
        const int64_t off = f.batch_counter[0] * f.b;
        f.idx1 += off * f.idx1_stride;
        f.idx2 += off * f.idx2_stride;
    }
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = i < f.b;
    const int64_t ii = live ? i : f.b - 1;
    int st = 0;
    int64_t r1, r2;
    split_rows<N>(f, ii, r1, r2, st);
    const bool bad = (st & sympa::ST_BAD_INDEX) != 0;
    constexpr int64_t ROW = 2 * N * N;
    if constexpr (MODEL == sympa::MODEL_UPPER) {
        // register-ordered form (siegel_math_bwd_split.hpp): planes leave one at a time; the factors wait in the LDS ([entry][lane]:
        // conflict-free) from the end of step 1 to the solves, Re G in the workspace slots of Hbar the lane has read by then
        const int lane = threadIdx.x;
        double* dtile = reinterpret_cast<double*>(lds);
        // entry k of my pack: (uniform base of entry k)[my pair] -- a scalar base and one 32-bit lane offset for all entries
        double* const ws = sa.ws;
        const int64_t wss = sa.ws_stride;
        const unsigned wo = (unsigned)ii * 8u;         // byte offset of my pair inside an entry (the workspace is < 4 GB per entry)
        const bool on = live && !bad;
        constexpr int TRI = N * (N + 1) / 2;
        sympa::CMat<N> z1, z2;
        gather_pair_passes<N, false>(f.base1, (int)r1, f.base2, (int)r2, lds, z1, z2);
        wave_lds_fence();
        sympa::pair_adjoint_gradient_upper<N>(
            z1, z2, [&](const int k) { return *ws_at(ws + k * wss, wo); },
            [&](const int which, const sympa::Tri<N, false>& l) {
                double* p = dtile + which * TRI * 64 + lane;
SYMPA_UNROLL
                for (int r = 0; r < N; ++r) {
                    p[sympa::tri_index(N, r, r) * 64] = l.rdiag[r];
SYMPA_UNROLL
                    for (int c = 0; c < r; ++c) p[sympa::tri_index(N, c, r) * 64] = l.re[r][c];
                }
            },
            [&](const int which, sympa::Tri<N, false>& l) {
                const double* p = dtile + which * TRI * 64 + lane;
SYMPA_UNROLL
                for (int r = 0; r < N; ++r) {
                    l.rdiag[r] = p[sympa::tri_index(N, r, r) * 64];
SYMPA_UNROLL
                    for (int c = 0; c < r; ++c) l.re[r][c] = p[sympa::tri_index(N, c, r) * 64];
                }
            },
            [&](const int k, const double g) { *ws_at(ws + (P::H_RE + k) * wss, wo) = g; },
            [&](const int k) { return *ws_at(ws + (P::H_RE + k) * wss, wo); },
            [&](const double (&m)[N][N]) { stage_plane<N>(m, dtile, on); },
            [&](const int point, const int plane, const double sign) {
                // nothing moves across this point: the scheduler otherwise pulls the first planes' atomics in front of the last
                // congruences, whose spilled operands (scratch loads) then wait for every atomic before them
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SCATTER) {
                    flush_plane_atomic<N>(sign, (int)(point == 0 ? r1 : r2), (point == 0 ? a.g1 : a.g2) + plane * N * N, dtile);
                } else {
                    const int64_t wave_first = i - lane;
                    const int64_t left = f.b - wave_first;
                    flush_plane_rows<N>(sign, (point == 0 ? a.g1 : a.g2) + wave_first * ROW + plane * N * N, dtile,
                                        left >= 64 ? 64 : (left > 0 ? (int)left : 0));
                }
            });
        return;
    } else {
    sympa::CMat<N> g1, g2;
    {
        sympa::CMat<N> z1, z2;
        gather_pair_passes<N, false>(f.base1, (int)r1, f.base2, (int)r2, lds, z1, z2);
        double pack[P::LEN];
        const unsigned wo = (unsigned)ii * 8u;
SYMPA_UNROLL
        for (int k = 0; k < P::LEN; ++k) pack[k] = *ws_at(sa.ws + k * sa.ws_stride, wo);
        sympa::pair_adjoint_gradient<N, MODEL>(z1, z2, pack, g1, g2);
    }
    if constexpr (SCATTER) {
        double* dtile = reinterpret_cast<double*>(lds);
        if constexpr (N >= 7) {
            scatter_add_rows_outlined<N>(g1, (int)r1, a.g1, dtile, live && !bad);
            scatter_add_rows_outlined<N>(g2, (int)r2, a.g2, dtile, live && !bad);
        } else {
            scatter_add_rows<N>(g1, (int)r1, a.g1, dtile, live && !bad);
            scatter_add_rows<N>(g2, (int)r2, a.g2, dtile, live && !bad);
        }
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
    }
}

template <int N, int MODEL>
int launch_bwd_split_spectral(const SplitArgs& sa, hipStream_t s) {
    const unsigned grid = (unsigned)((sa.a.f.b + 63) / 64);
    hipLaunchKernelGGL((siegel_bwd_spectral_kernel<N, MODEL>), dim3(grid), dim3(64), 0, s, sa);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

template <int N, int MODEL, bool SCATTER>
int launch_bwd_split_gradient(const SplitArgs& sa, hipStream_t s) {
    const unsigned grid = (unsigned)((sa.a.f.b + 63) / 64);
    hipLaunchKernelGGL((siegel_bwd_gradient_kernel<N, MODEL, SCATTER>), dim3(grid), dim3(64), 0, s, sa);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

// siegel_bwd_split.hip: workspace size and dispatch (one translation unit per kernel: siegel_bwd_split_*_*.hip)
int64_t bwd_split_workspace_bytes(int64_t b, int n, int model);
bool bwd_split_available(int n, int model);
int launch_bwd_split(const BwdArgs& a, int n, int model, bool scatter, void* workspace, int64_t workspace_bytes, hipStream_t s);
// siegel_bwd.hip: the one-stage, one-pair-per-lane kernels of dims 1..8 (siegel_bwd_kernel), whatever the default dispatch prefers;
// with a.chunk_flags (dims 5..8) only the flagged 64-pair chunks are processed, by a small fixed grid
int launch_bwd_one_lane(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s);

}  // namespace sympa_hip
