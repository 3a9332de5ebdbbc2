// Round 5: what does one global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) cost a CU, and does it depend on how many lanes are active?
// Every wave issues REPS x 16 DMA instructions into its own 16 KB LDS region from a small (L2-resident) buffer and waits once per 16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lds_dma_rate lds_dma_rate.hip && ./lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void* glb_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int ACTIVE>
__global__ __launch_bounds__(64) void dma_kernel(const double* __restrict__ src, int reps, int rows, double* out) {
    extern __shared__ double lds[];                    // 16 KB per block
    const int lane = threadIdx.x;
    unsigned r = blockIdx.x * 2654435761u;
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            r = r * 1664525u + 1013904223u;
            const unsigned row = __builtin_amdgcn_readfirstlane(r >> 8) % (unsigned)rows;
            const double* p = src + (size_t)row * 128 + 2 * lane;          // 1 KB rows
            if (lane < ACTIVE) __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(lds + j * 128), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0)
    }
    __syncthreads();
    if (out != nullptr && lds[lane] == 1.2345) out[blockIdx.x] = lds[lane];
}

template <int ACTIVE>
double run(const double* src, int rows, int blocks, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(dma_kernel<ACTIVE>, dim3(blocks), dim3(64), 16384, 0, src, 4, rows, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(dma_kernel<ACTIVE>, dim3(blocks), dim3(64), 16384, 0, src, reps, rows, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    for (int rows : {64, 4096, 65536}) {               // 64 KB (L1/L2), 4 MB (L2), 64 MB (MALL) of 1 KB rows
        double* src;
        hipMalloc(&src, (size_t)rows * 1024);
        hipMemset(src, 0, (size_t)rows * 1024);
        for (int per_cu : {1, 2, 4, 8}) {
            const int blocks = cus * per_cu, reps = 256;
            const double t64 = run<64>(src, rows, blocks, reps), t54 = run<54>(src, rows, blocks, reps), t36 = run<36>(src, rows, blocks, reps),
                         t16 = run<16>(src, rows, blocks, reps);
            const double instr_per_cu = (double)per_cu * reps * 16;
            auto cyc = [&](double us) { return us * 1e-6 * ghz * 1e9 / instr_per_cu; };
            printf("table %6d KB  %d waves/CU  us: %8.1f %8.1f %8.1f %8.1f   cycles per DMA instruction and CU at %4.2f GHz (64/54/36/16 lanes): %6.1f %6.1f %6.1f %6.1f   "
                   "GB/s chip (64 lanes): %7.0f\n", rows, per_cu, t64, t54, t36, t16, ghz, cyc(t64), cyc(t54), cyc(t36), cyc(t16),
                   (double)blocks * reps * 16 * 1024 / (t64 * 1e-6) * 1e-9);
        }
        hipFree(src);
    }
    return 0;
}
