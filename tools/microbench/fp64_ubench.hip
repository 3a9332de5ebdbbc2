// fp64 VALU micro-benchmarks for gfx950 (evidence for DESIGN.md section 5):
//   * accuracy of v_rsq_f64 / v_rcp_f64 / v_sqrt_f64 seeds (max relative error)
//   * cycles per wave-instruction of v_fma_f64 (dependent chain vs 8 independent chains),
//     v_mul/v_add, v_rsq_f64, v_rcp_f64 with 1 wave per SIMD (s_memtime stamps)
//   * chip-level fp64 FMA throughput at 1, 2, 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

__global__ void seeds(const double* x, double* rsq, double* rcp, double* sq, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        rsq[i] = __builtin_amdgcn_rsq(x[i]);
        rcp[i] = __builtin_amdgcn_rcp(x[i]);
        sq[i] = __builtin_amdgcn_sqrt(x[i]);
    }
}

template <int MODE>
__global__ void lat(double* out, unsigned long long* cyc, int iters) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    const double m = 0.999999, c = 1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // one dependent FMA chain, 8 per iteration
#pragma unroll
            for (int k = 0; k < 8; ++k) a[0] = __builtin_fma(a[0], m, c);
        } else if (MODE == 1) {   // 8 independent FMA chains
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], m, c);
        } else if (MODE == 2) {   // 8 independent rsq
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_amdgcn_rsq(a[k]) + 1.0;
        } else if (MODE == 3) {   // 8 independent rcp
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_amdgcn_rcp(a[k]) + 1.0;
        } else if (MODE == 4) {   // 8 independent mul
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = a[k] * m;
        } else if (MODE == 5) {   // 8 independent add
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = a[k] + c;
        } else if (MODE == 6) {   // dependent rsq chain
#pragma unroll
            for (int k = 0; k < 8; ++k) a[0] = __builtin_amdgcn_rsq(a[0]);
        } else if (MODE == 7) {   // 8 independent IEEE sqrt (library)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_sqrt(a[k]) + 1.0;
        } else if (MODE == 8) {   // 8 independent IEEE div
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = 1.0 / a[k] + 1.0;
        } else if (MODE == 9) {   // 8 independent f32 fma for comparison
            float f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = (float)a[k];
#pragma unroll
            for (int r = 0; r < 1; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = __builtin_fmaf(f[k], 0.99999f, 1e-7f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = f[k];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// EXEC-masked variant: only the first `active` lanes of each wave run the FMA loop
__global__ void lat_masked(double* out, unsigned long long* cyc, int iters, int active) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    const double m = 0.999999, c = 1e-7;
    unsigned long long t0 = 0, t1 = 0;
    if ((threadIdx.x & 63) < active) {
        t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], m, c);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void thr_masked(double* out, int iters, int active) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    const double m = 0.999999, c = 1e-7;
    if ((threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], m, c);
        }
    }
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void thr(double* out, int iters) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    const double m = 0.999999, c = 1e-7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(a[k], m, c);
    }
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> hx(n);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> ex(-20, 20);
    for (auto& v : hx) v = std::exp(ex(rng));
    double *x, *a, *b, *c;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8));
    CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
    seeds<<<n / 256, 256>>>(x, a, b, c, n);
    std::vector<double> ha(n), hb(n), hc(n);
    CK(hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        e1 = std::fmax(e1, std::fabs(ha[i] * std::sqrt(hx[i]) - 1.0));
        e2 = std::fmax(e2, std::fabs(hb[i] * hx[i] - 1.0));
        e3 = std::fmax(e3, std::fabs(hc[i] / std::sqrt(hx[i]) - 1.0));
    }
    printf("seed max rel err: v_rsq_f64 %.3e  v_rcp_f64 %.3e  v_sqrt_f64 %.3e\n", e1, e2, e3);

    unsigned long long* cyc;
    CK(hipMalloc(&cyc, 8));
    const int iters = 4096;
    const char* names[] = {"fma dep chain", "fma 8 indep", "rsq 8 indep(+add)", "rcp 8 indep(+add)", "mul 8 indep",
                           "add 8 indep", "rsq dep chain", "IEEE sqrt 8 indep(+add)", "IEEE div 8 indep(+add)", "cvt+fmaf"};
#define RUN(M) { lat<M><<<1, 64>>>(a, cyc, iters); unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); \
                 printf("1 wave: %-26s %7.2f cycles per op\n", names[M], (double)h / (iters * 8.0)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    for (int act : {64, 32, 16}) {
        lat_masked<<<1, 64>>>(a, cyc, iters, act);
        unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("1 wave, %2d active lanes: fma 8 indep %7.2f cycles per op\n", act, (double)h / (iters * 8.0));
    }
    {
        hipEvent_t s0, e0; CK(hipEventCreate(&s0)); CK(hipEventCreate(&e0));
        for (int act : {64, 32}) for (int wps : {1, 2, 4}) {
            int blocks = 256 * wps; const int it2 = 20000;
            thr_masked<<<blocks, 256>>>(a, 100, act);
            CK(hipEventRecord(s0)); thr_masked<<<blocks, 256>>>(a, it2, act); CK(hipEventRecord(e0)); CK(hipEventSynchronize(e0));
            float ms; CK(hipEventElapsedTime(&ms, s0, e0));
            printf("chip, %d active lanes, %d wave(s)/SIMD: %.3f ms  -> %.1f G lane-FMA/s\n", act, wps, ms, 8.0 * it2 * act * 4.0 * blocks / ms / 1e6);
        }
    }
    // chip throughput: waves per SIMD = 1, 2, 4, 8
    hipEvent_t s, e;
    CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;   // 256 threads = 4 waves = 1 per SIMD per block
        thr<<<blocks, 256>>>(a, 100);
        CK(hipEventRecord(s));
        const int it2 = 20000;
        thr<<<blocks, 256>>>(a, it2);
        CK(hipEventRecord(e));
        CK(hipEventSynchronize(e));
        float ms;
        CK(hipEventElapsedTime(&ms, s, e));
        double flops = 2.0 * 8 * it2 * 256.0 * blocks;
        printf("chip fp64 FMA, %d wave(s)/SIMD: %.1f TFLOP/s  (%.3f ms)\n", wps, flops / ms / 1e9, ms);
    }
    return 0;
}
