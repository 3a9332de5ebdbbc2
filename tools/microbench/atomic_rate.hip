// fp64 atomic-add throughput into a table of 1 KB rows (the scatter of the Siegel backward at n = 8), many waves per SIMD:
//   mode 0: 64 lanes -> 64 consecutive doubles of one plane of a random row (4 lines of 128 B per instruction)
//   mode 1: lanes 0..35 only (the upper triangle's worth of adds, still 4 lines)
//   mode 2: 64 lanes -> 36 consecutive doubles of one packed row + 28 of the next (packed triangles, 576 B rows)
//   mode 3: as 0 with sc1 (device scope) set explicitly through atomicAdd
// hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void gadd(double* p, double v) {
    typedef __attribute__((address_space(1))) double gdouble;
    (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble*)p, v);
}

template <int MODE>
__global__ void k(double* table, const int* rows, int planes_per_wave, int num_rows) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int* r = rows + (long)wave * planes_per_wave;
    for (int t = 0; t < planes_per_wave; ++t) {
        const int row = r[t];
        if (MODE == 0) gadd(table + (long)row * 128 + (t & 1) * 64 + lane, 1.0);
        if (MODE == 1) { if (lane < 36) gadd(table + (long)row * 128 + (t & 1) * 64 + lane, 1.0); }
        if (MODE == 2) {
            const int row2 = (lane < 36) ? row : (row + 1 < num_rows ? row + 1 : 0);
            const int e = (lane < 36) ? lane : lane - 36;
            gadd(table + (long)row2 * 72 + (t & 1) * 36 + e, 1.0);
        }
        if (MODE == 3) atomicAdd(table + (long)row * 128 + (t & 1) * 64 + lane, 1.0);
    }
}

int main() {
    const int num_rows = 45500, pairs = 262144, planes = pairs * 4;      // 4 planes per pair, one instruction each (modes 0, 1, 3)
    const int waves = 4096 * 4, ppw = planes / waves;
    double* table; int* rows;
    hipMalloc(&table, (size_t)num_rows * 128 * 8);
    hipMemset(table, 0, (size_t)num_rows * 128 * 8);
    std::vector<int> h(planes);
    srand(1);
    for (int i = 0; i < planes; ++i) h[i] = rand() % num_rows;
    hipMalloc(&rows, planes * sizeof(int));
    hipMemcpy(rows, h.data(), planes * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int mode, const char* name) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, table, rows, ppw, num_rows);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, table, rows, ppw, num_rows);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(waves / 4), dim3(256), 0, 0, table, rows, ppw * 36 / 64, num_rows);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(waves / 4), dim3(256), 0, 0, table, rows, ppw, num_rows);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-58s %8.1f us\n", name, best * 1000);
    };
    run(0, "full planes, 64 lanes x 8 B contiguous (4 lines)");
    run(1, "36 of 64 lanes active (4 lines, 36 adds)");
    run(2, "packed triangles: 36/64 of the instructions, ~5 lines each");
    run(3, "full planes through atomicAdd (device scope)");
    return 0;
}
