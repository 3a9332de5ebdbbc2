// Round 5: what does a second (third) resident wave per SIMD buy the dims-8 eigenvalue stage (Householder + lockstep QL + log1p, the
// 5 k instructions that are half of the packed forward)?  Same code, same registers; the occupancy is set by the LDS a block asks for.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../sympa_amd/csrc -o eigen_occupancy eigen_occupancy.hip && ./eigen_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include "siegel_math.hpp"

template <int N>
__global__ __launch_bounds__(64, 3) void eig_kernel(double* out, int reps, double seed) {
    extern __shared__ double dyn[];
    const int lane = threadIdx.x;
    const unsigned id = blockIdx.x * 64 + lane;
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        sympa::Herm<N> h;
        // a positive definite Hermitian matrix that differs per lane and repetition (E^H E of a pseudo-random E)
        sympa::CMat<N> e;
        unsigned s = id * 2654435761u + r * 40503u + 12345u;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                s = s * 1664525u + 1013904223u;
                e.re[i][j] = (double)(s >> 8) * (1.0 / 16777216.0) - 0.5 + (i == j ? seed : 0.0);
                s = s * 1664525u + 1013904223u;
                e.im[i][j] = (double)(s >> 8) * (1.0 / 16777216.0) - 0.5;
            }
        sympa::gram<N>(e, h);
        int st = 0;
        acc += sympa::distance_from_h<N, sympa::MODEL_UPPER>(h, true, 0, nullptr, 1e5, nullptr, st);
    }
    if (dyn[lane] == 123.456) acc += 1.0;      // keeps the dynamic LDS allocation alive
    out[id] = acc;
}

int main() {
    constexpr int N = 8;
    const int waves = 4096, reps = 4;
    double* out;
    hipMalloc(&out, waves * 64 * sizeof(double));
    hipFuncSetAttribute(reinterpret_cast<const void*>(eig_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(eig_kernel<N>));
    printf("eigen stage + Gram, n = %d: %d registers, %zu bytes scratch; %d waves x %d repetitions\n", N, fa.numRegs, (size_t)fa.localSizeBytes, waves, reps);
    const size_t lds[] = {40 * 1024, 20 * 1024, 13 * 1024, 8 * 1024};       // 1, 2, 3, (3: register-limited) waves per SIMD
    for (int pass = 0; pass < 2; ++pass)
        for (size_t l : lds) {
            int per_cu = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, eig_kernel<N>, 64, l);
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(eig_kernel<N>, dim3(waves), dim3(64), l, 0, out, reps, 3.0);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(eig_kernel<N>, dim3(waves), dim3(64), l, 0, out, reps, 3.0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("  LDS %5zu B per block -> %d blocks per CU (%.2f waves per SIMD): %8.1f us per launch, %6.2f us per wave-repetition and SIMD slot\n",
                   l, per_cu, per_cu / 4.0, ms * 1e3 / 5, ms * 1e3 / 5 / (waves / 1024.0 * reps));
        }
    return 0;
}
