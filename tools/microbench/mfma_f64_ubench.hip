// fp64 MFMA micro-benchmarks for gfx950 (evidence for DESIGN.md: is the matrix pipe worth using for the
// 16 x 16 spd / Siegel kernels, which are bound by fp64 VALU issue?):
//   * cycles per v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 for one wave (dependent accumulator vs
//     four independent accumulators), s_memtime stamps
//   * the same with N independent v_fma_f64 between two MFMAs (do VALU and matrix pipe overlap inside ONE wave?)
//   * chip-level FLOP/s at 1, 2, 4 waves per SIMD, MFMA alone, VALU alone, and an MFMA wave beside a VALU wave
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_ubench mfma_f64_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x4, one dependent accumulator      MODE 1: 16x16x4, four independent accumulators
// MODE 2: 4x4x4 (4 blocks), dependent             MODE 3: 4x4x4, four independent accumulators
// VALU = number of independent v_fma_f64 issued after every MFMA by the same wave
template <int MODE, int VALU>
__global__ void lat(double* out, unsigned long long* cyc, int iters) {
    const double a = 1.0 + 1e-3 * threadIdx.x, b = 1.0 - 1e-3 * threadIdx.x;
    v4d c[4];
    double s[4];
    double f[8];
    for (int k = 0; k < 4; ++k) { c[k] = (v4d){0.0, 0.0, 0.0, 0.0}; s[k] = 0.0; }
    for (int k = 0; k < 8; ++k) f[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    const double m = 0.999999, cc = 1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 0) c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[0], 0, 0, 0);
            if (MODE == 1) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
            if (MODE == 2) s[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[0], 0, 0, 0);
            if (MODE == 3) s[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[k], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU; ++v) f[v % 8] = __builtin_fma(f[v % 8], m, cc);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
    for (int k = 0; k < 4; ++k) r += c[k].x + c[k].y + c[k].z + c[k].w + s[k];
    for (int k = 0; k < 8; ++k) r += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// chip throughput.  KIND 0: MFMA 16x16x4 only, 1: VALU FMA only, 2: even waves MFMA / odd waves VALU,
// 3: every wave interleaves 1 MFMA + 8 VALU FMA
template <int KIND>
__global__ void thr(double* out, int iters) {
    const double a = 1.0 + 1e-3 * threadIdx.x, b = 1.0 - 1e-3 * threadIdx.x;
    v4d c[4];
    double f[8];
    for (int k = 0; k < 4; ++k) c[k] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < 8; ++k) f[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    const double m = 0.999999, cc = 1e-7;
    const bool mfma_wave = (KIND == 0) || (KIND == 2 && (((threadIdx.x >> 6) + blockIdx.x) & 1) == 0);   // wave-uniform
    if (KIND == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 8; ++v) f[v] = __builtin_fma(f[v], m, cc);
            }
        }
    } else if (mfma_wave) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = __builtin_fma(f[k], m, cc);
        }
    }
    double r = 0;
    for (int k = 0; k < 4; ++k) r += c[k].x + c[k].y + c[k].z + c[k].w;
    for (int k = 0; k < 8; ++k) r += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    double* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 8 * 256 * 256 * 8));
    CK(hipMalloc(&cyc, 8));
    const int iters = 4096;
#define RUN(M, V, NAME) { lat<M, V><<<1, 64>>>(out, cyc, iters); unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); \
        printf("1 wave: %-52s %7.2f cycles per MFMA (+%d v_fma_f64 each)\n", NAME, (double)h / (iters * 4.0), V); }
    RUN(0, 0, "v_mfma_f64_16x16x4_f64 dependent accumulator")
    RUN(1, 0, "v_mfma_f64_16x16x4_f64 4 independent accumulators")
    RUN(2, 0, "v_mfma_f64_4x4x4_4b_f64 dependent accumulator")
    RUN(3, 0, "v_mfma_f64_4x4x4_4b_f64 4 independent accumulators")
    RUN(1, 2, "16x16x4 indep + 2 v_fma_f64")
    RUN(1, 4, "16x16x4 indep + 4 v_fma_f64")
    RUN(1, 6, "16x16x4 indep + 6 v_fma_f64")
    RUN(1, 8, "16x16x4 indep + 8 v_fma_f64")
    RUN(1, 12, "16x16x4 indep + 12 v_fma_f64")
    RUN(3, 2, "4x4x4 indep + 2 v_fma_f64")
    RUN(3, 4, "4x4x4 indep + 4 v_fma_f64")
    hipEvent_t s, e;
    CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    const int it2 = 4000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        float ms;
        thr<0><<<blocks, 256>>>(out, 50);
        CK(hipEventRecord(s)); thr<0><<<blocks, 256>>>(out, it2); CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
        CK(hipEventElapsedTime(&ms, s, e));
        // one 16x16x4 MFMA = 16*16*4*2 = 2048 flop per wave-instruction
        printf("chip, %d wave(s)/SIMD, MFMA f64 16x16x4 only : %7.1f TFLOP/s (%.3f ms)\n", wps, 2048.0 * 4 * it2 * 4.0 * blocks / ms / 1e9, ms);
        thr<1><<<blocks, 256>>>(out, 50);
        CK(hipEventRecord(s)); thr<1><<<blocks, 256>>>(out, it2); CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
        CK(hipEventElapsedTime(&ms, s, e));
        printf("chip, %d wave(s)/SIMD, v_fma_f64 only            : %7.1f TFLOP/s (%.3f ms)\n", wps, 128.0 * 32 * it2 * 4.0 * blocks / ms / 1e9, ms);
        if (wps >= 2) {
            thr<2><<<blocks, 256>>>(out, 50);
            CK(hipEventRecord(s)); thr<2><<<blocks, 256>>>(out, it2); CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
            CK(hipEventElapsedTime(&ms, s, e));
            const double fl = (2048.0 * 4 + 128.0 * 32) * it2 * 2.0 * blocks;   // half the waves each
            printf("chip, %d wave(s)/SIMD, half MFMA waves + half VALU waves: %7.1f TFLOP/s combined (%.3f ms)\n", wps, fl / ms / 1e9, ms);
        }
        thr<3><<<blocks, 256>>>(out, 50);
        CK(hipEventRecord(s)); thr<3><<<blocks, 256>>>(out, it2); CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
        CK(hipEventElapsedTime(&ms, s, e));
        printf("chip, %d wave(s)/SIMD, every wave 1 MFMA + 8 v_fma_f64  : %7.1f TFLOP/s combined (%.3f ms)\n", wps,
               (2048.0 + 128.0 * 8) * 4 * it2 * 4.0 * blocks / ms / 1e9, ms);
    }
    return 0;
}
