// Shared host/device declarations of the translation units of libsympa_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/sympa_hip.h"
#include "siegel_math.hpp"
#include "siegel_gather.hpp"

namespace sympa_hip {

constexpr int BLOCK = 256;
// Forward kernels, dims >= SYMPA_FWD_WAVE_BLOCK_FROM: a block is ONE wave.  These kernels hold 390-512 registers (one wave per
// SIMD) and their waves never synchronise with each other; with four-wave blocks a CU takes its next block only when the
// slowest of the four waves (gather latency, lockstep QL sweeps) is through -- with one-wave blocks every SIMD takes its
// next wave by itself.  Measured (profiles/r04_fwd_wave_blocks.txt, 262 144 pairs, same box, interleaved): upper n = 8 152.5 -> 149.8 us,
// bounded n = 8 240.2 -> 237.7, upper n = 7 106.5 -> 105.5, dims 5, 6 -0.1..-0.7 %: small, never slower, bit-identical results.
constexpr int SYMPA_FWD_WAVE_BLOCK_FROM = 5;
constexpr int SYMPA_INTERNAL_FLAG_STAGGER = 0x4000;      // set by the forward launcher, never by callers (siegel_dist_kernel.hpp)
constexpr int fwd_block(int n) { return n >= SYMPA_FWD_WAVE_BLOCK_FROM ? 64 : BLOCK; }

struct DistArgs {
    const double* base1;   // z1 [b,2,n,n]   or the table
    const double* base2;   // z2 [b,2,n,n]   or the table
    const int64_t* idx1;   // nullptr -> row i
    const int64_t* idx2;
    int64_t idx1_stride;
    int64_t idx2_stride;
    int64_t num_rows;      // rows addressable through idx (bounds check)
    int64_t b;
    const double* metric_w;
    const double* scale;   // device scalar or nullptr
    double inv_scale_coef;
    double inv_eps;        // 1 / eps, computed on the host
    double* out;
    double* vvd;
    int32_t* status;
    int metric;
    int flags;
    int64_t ap_cols;       // > 0: all-pairs mode, pair p = (ap_row0 + p / ap_cols, p % ap_cols), idx1/idx2 unused
    int64_t ap_row0;
    int ap_sym;            // all-pairs mode over the FULL matrix using d(i, j) = d(j, i): pair p runs over the
                           // ap_cols (ap_cols + 1) / 2 pairs i <= j in row-major triangular order and is stored twice
    const int64_t* batch_counter;   // training graph (backward, one pair per lane): device word c -> this launch processes
                                    // pairs [c b, (c + 1) b) of idx1 / idx2 / graph_dist (an epoch's triplets stay in one buffer
                                    // and the step kernel increments c: no per-step copy of the batch); null: off
};

#if defined(__HIPCC__)
// All-pairs mode: rows of pair p and the place(s) its value goes to.
__device__ __forceinline__ void ap_pair(const DistArgs& a, const int64_t p, int64_t& r1, int64_t& r2) {
    if (!a.ap_sym) {
        r1 = a.ap_row0 + p / a.ap_cols;
        r2 = p % a.ap_cols;
        return;
    }
    // row i of the upper triangle starts at off(i) = i N - i (i - 1) / 2:  i = floor((2N + 1 - sqrt((2N + 1)^2 - 8 p)) / 2)
    const int64_t n = a.ap_cols;
    const double t = (double)(2 * n + 1);
    int64_t i = (int64_t)((t - sympa::d_sqrt(t * t - 8.0 * (double)p)) * 0.5);
    i = i < 0 ? 0 : (i >= n ? n - 1 : i);
    if ((i + 1) * n - (i + 1) * i / 2 <= p) ++i;          // the square root is good to an ulp: at most one step off
    if (i * n - i * (i - 1) / 2 > p) --i;
    r1 = i;
    r2 = i + (p - (i * n - i * (i - 1) / 2));
}
__device__ __forceinline__ void ap_store(const DistArgs& a, const int64_t p, const int64_t r1, const int64_t r2, const double d) {
    if (!a.ap_sym) {
        __builtin_nontemporal_store(d, a.out + p);
        return;
    }
    __builtin_nontemporal_store(d, a.out + r1 * a.ap_cols + r2);
    if (r1 != r2) __builtin_nontemporal_store(d, a.out + r2 * a.ap_cols + r1);
}
#endif

// Arguments of the PERSISTENT forward kernels (siegel_packed_kernel.hpp): up to SYMPA_MAX_FUSED_BATCHES batches per launch, tiles
// of 64 pairs numbered through the batches; every wave walks tiles t, t + grid, ...
struct PackedArgs {
    const double* pack;                               // the packed table (packed kernels) / the first points' rows (dense kernel)
    const double* base2;                              // dense kernel: the second points' rows (the same table, or z2 of a dist call)
    int identity;                                     // dense kernel: pair i reads rows (i, i) of (pack, base2): pre-gathered points
    int64_t num_rows;
    const int64_t* idx1[SYMPA_MAX_FUSED_BATCHES];     // src ids of batch k, element i at idx1[k][i * stride1]
    const int64_t* idx2[SYMPA_MAX_FUSED_BATCHES];
    double* out[SYMPA_MAX_FUSED_BATCHES];
    int64_t b[SYMPA_MAX_FUSED_BATCHES];
    unsigned tile_end[SYMPA_MAX_FUSED_BATCHES];       // exclusive prefix end of batch k, in tiles of 64 pairs
    int64_t stride1, stride2;
    const double* metric_w;
    const double* scale;
    double inv_scale_coef, inv_eps;
    int32_t* status;
    int metric;
    int num_batches;
    unsigned tiles;
    int stagger;                                      // first round: CU j of every XCD starts j x 0.6 us late (tables beyond the L2s)
};

// siegel_packed.hip: the persistent DENSE forward of the upper model, dims 7, 8 (dense_forward_kernel): fills tile_end / tiles /
// stagger from the batches already set in `a` (idx1 / idx2 / out / b for num_batches entries) and launches
int launch_dense_persistent(PackedArgs& a, int n, hipStream_t s);

// Registry of kernel instantiations of the inline-asm DPP layouts (sixteen / eight lanes per pair or row) that a numerical
// self-check found to disagree with the one-lane kernels (C-ABI sympa_set_instance_fallback; sympa_amd/selfcheck.py runs the
// check on first use of every instantiation): the dispatchers route those to the one-lane kernels.  family = SYMPA_FAMILY_*.
bool instance_fallback(int family, int model, int n);

// thread-local message behind sympa_last_error()
char* last_error_buffer();
int fail(int code, const char* msg);
int validate(const DistArgs& a, int model, int n);

// siegel_coop.hip: 9 <= n <= 16, sixteen lanes per pair
int launch_siegel_coop(const DistArgs& a, int n, int model, hipStream_t s);
// siegel_coop_half.hip: dims 7, 8 with eight lanes per pair (SYMPA_FLAG_COOP; A/B only)
int launch_siegel_coop_half(const DistArgs& a, int n, int model, hipStream_t s);

}  // namespace sympa_hip
