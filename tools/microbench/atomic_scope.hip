// fp64 atomic-add throughput on gfx950 by memory scope and destination layout (evidence for DESIGN.md section 9):
//   row-shaped scatter like the backward kernel (each wave instruction adds 64 consecutive doubles = 2 rows of
//   32 doubles picked by index), 131 072 row updates over a 5 041-row table per launch.
//   MODE 0: agent scope, one shared table            (what atomicAdd() compiles to)
//   MODE 1: workgroup scope, one table PER XCD       (atomic resolved in the XCD's own L2; XCC_ID hardware register)
//   MODE 2: agent scope, one table per XCD           (separates the effect of contention from the effect of scope)
//   MODE 3: plain stores of the same shape           (upper bound: no read-modify-write)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(256) void scatter(const int* __restrict__ rows, double* __restrict__ table, long table_doubles,
                                               int updates, int* __restrict__ xcc_seen) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int x = xcc_id();
    if (lane == 0 && xcc_seen) xcc_seen[x & 15] = 1;
    double* base = (MODE == 1 || MODE == 2) ? table + (long)(x & 7) * table_doubles : table;
    // each wave: 64 row updates = 32 instructions of 2 rows
    const int first = wave * 64;
    if (first >= updates) return;
    const int myrow = rows[first + lane];
#pragma unroll 4
    for (int t = 0; t < 32; ++t) {
        const int p = 2 * t + (lane >> 5), e = lane & 31;
        const int r = __shfl(myrow, p);
        double* dst = base + (long)r * 32 + e;
        const double val = 1.0 + e;
        if (MODE == 0 || MODE == 2) __hip_atomic_fetch_add(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) __hip_atomic_fetch_add(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 3) *dst = val;
        else if (MODE == 4) {       // only the 20 upper-triangle elements of each [2,4,4] row, other lanes idle
            const int ee = e & 15, i = ee >> 2, j = ee & 3;
            if (i <= j) __hip_atomic_fetch_add(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 5) {     // half of the rows only (every other update skipped): time vs number of updates
            if (!(p & 1)) __hip_atomic_fetch_add(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 7) {     // loop overhead only: no memory operation inside the loop
            asm volatile("" :: "v"(dst));
        } else if (MODE == 8) {     // 16 bytes per lane: dwordx4 plain stores (two doubles), half the instructions
            if (!(t & 1)) { double2 v2 = {val, val}; *reinterpret_cast<double2*>(base + (long)r * 32 + ((2 * e) & 31)) = v2; }
        } else if (MODE == 6) {     // no contention: update u goes to its own row (table of `updates` rows needed)
            double* d2 = table + (long)(first + p) * 32 + e;
            __hip_atomic_fetch_add(d2, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// packed rows of W doubles (W = 20: upper triangles of [2,4,4]; W = 24: the same padded to 192 B): the wave's 64 row
// updates are W*64 doubles, issued as W instructions of 64 consecutive (update, element) slots
template <int W>
__global__ __launch_bounds__(256) void scatter_packed(const int* __restrict__ rows, double* __restrict__ table, int updates) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int first = wave * 64;
    if (first >= updates) return;
    const int myrow = rows[first + lane];
#pragma unroll 4
    for (int t = 0; t < W; ++t) {
        const int slot = t * 64 + lane;
        const int p = slot / W, e = slot - p * W;
        const int r = __shfl(myrow, p);
        __hip_atomic_fetch_add(table + (long)r * W + e, 1.0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int W>
int run_packed(const char* name, const int* d_rows, double* d_table, int updates) {
    const int waves = updates / 64, blocks = waves / 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(scatter_packed<W>, dim3(blocks), dim3(256), 0, 0, d_rows, d_table, updates);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(scatter_packed<W>, dim3(blocks), dim3(256), 0, 0, d_rows, d_table, updates);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %8.2f us  (%d doubles per row update)\n", name, ms * 1e3 / reps, W);
    return 0;
}

template <int MODE>
int run(const char* name, const int* d_rows, double* d_table, long table_doubles, int updates, int* d_seen) {
    const int waves = updates / 64, blocks = waves / 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(d_table, 0, 8 * table_doubles * sizeof(double)));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(scatter<MODE>, dim3(blocks), dim3(256), 0, 0, d_rows, d_table, table_doubles, updates, d_seen);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(scatter<MODE>, dim3(blocks), dim3(256), 0, 0, d_rows, d_table, table_doubles, updates, d_seen);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, bytes = (double)updates * 256;
    // checksum: total added over all copies must equal (3 + reps) * updates * sum(1..32) for the atomic modes
    std::vector<double> h(8 * table_doubles);
    CK(hipMemcpy(h.data(), d_table, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    double s = 0; for (double v : h) s += v;
    const double want = (double)(3 + reps) * updates * (32.0 * 33.0 / 2.0);
    printf("%-44s %8.2f us  %7.1f GB/s of updates   checksum %s\n", name, us, bytes / us * 1e-3,
           MODE >= 3 ? "n/a" : (s == want ? "ok" : "MISMATCH"));
    return 0;
}

int main() {
    const int nrows = 5041, updates = 131072;
    const long table_doubles = (long)nrows * 32;
    std::vector<int> rows(updates);
    std::mt19937 g(1);
    for (auto& r : rows) r = g() % nrows;
    int* d_rows; double* d_table; int* d_seen;
    CK(hipMalloc(&d_rows, updates * sizeof(int)));
    CK(hipMalloc(&d_table, (8 * table_doubles + (long)updates * 32) * sizeof(double)));
    CK(hipMalloc(&d_seen, 16 * sizeof(int)));
    CK(hipMemset(d_seen, 0, 16 * sizeof(int)));
    CK(hipMemcpy(d_rows, rows.data(), updates * sizeof(int), hipMemcpyHostToDevice));
    if (run<0>("agent scope, shared table", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<2>("agent scope, table per XCD", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<1>("workgroup scope, table per XCD", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<3>("plain stores, table per XCD-less", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<4>("agent, shared, upper triangle only (20/32)", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<5>("agent, shared, every other update only", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<6>("agent, one private row per update (no reuse)", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<7>("loop overhead only (no memory op)", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<8>("plain dwordx4 stores, half the instructions", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run<0>("agent shared, 4x updates", d_rows, d_table, table_doubles, updates, d_seen)) return 1;
    if (run_packed<32>("packed kernel, 32 doubles per row (= dense)", d_rows, d_table, updates)) return 1;
    if (run_packed<24>("packed kernel, 24 doubles per row (192 B)", d_rows, d_table, updates)) return 1;
    if (run_packed<20>("packed kernel, 20 doubles per row (160 B)", d_rows, d_table, updates)) return 1;
    if (run_packed<16>("packed kernel, 16 doubles per row (128 B)", d_rows, d_table, updates)) return 1;
    int seen[16]; CK(hipMemcpy(seen, d_seen, sizeof(seen), hipMemcpyDeviceToHost));
    printf("XCC ids seen:"); for (int i = 0; i < 16; ++i) if (seen[i]) printf(" %d", i); printf("\n");
    return 0;
}
