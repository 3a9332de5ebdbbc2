// In-kernel timeline of the n=4 upper kernel (diagnostic build, never shipped): every wave stamps
// s_memrealtime (100 MHz) at stage boundaries; the host prints, relative to the earliest wave start of
// the launch, the mean/max time at which waves pass each boundary.  B = 65 536, 256 blocks x 256.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#include "../../sympa_amd/csrc/siegel_math.hpp"
#include "../../sympa_amd/csrc/siegel_gather.hpp"
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
using namespace sympa;
constexpr int NST = 8;

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

__global__ __launch_bounds__(256) void timeline(const double* table, const long long* idx, long long b, double* out,
                                                 unsigned long long* stamps) {
    __shared__ v2d lds[4 * DmaTile<4>::WAVE_SLOTS];
    constexpr int N = 4;
    unsigned long long t[NST];
    t[0] = stamp();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ii = i < b ? i : b - 1;
    long long r1 = idx[2 * ii], r2 = idx[2 * ii + 1];
    asm volatile("" :: "v"(r1), "v"(r2));
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    t[1] = stamp();
    CMat<N> z1, z2, e;
    v2d* tile = lds + (threadIdx.x >> 6) * DmaTile<4>::WAVE_SLOTS;
    gather_pair_dma_split<N>(table, (int)r1, table, (int)r2, tile, z1, z2);
    for (int a = 0; a < N; ++a) for (int c = a; c < N; ++c) asm volatile("" :: "v"(z1.re[a][c]), "v"(z1.im[a][c]), "v"(z2.re[a][c]), "v"(z2.im[a][c]));
    __builtin_amdgcn_sched_barrier(0);
    t[2] = stamp();
    Tri<N, false> l1, l2;
    bool ok = chol_real<N>(z1.im, l1);
    ok = chol_real<N>(z2.im, l2) && ok;
    for (int a = 0; a < N; ++a) for (int c = 0; c < N; ++c) { e.re[a][c] = z2.re[a][c] - z1.re[a][c]; e.im[a][c] = z2.im[a][c] - z1.im[a][c]; }
    solve_left<N, false>(l1, e);
    solve_right_t<N, false>(l2, e);
    Herm<N> h;
    gram<N>(e, h);
    asm volatile("" :: "v"(h.d[0]), "v"(h.re[0][1]));
    __builtin_amdgcn_sched_barrier(0);
    t[3] = stamp();
    const bool conv = herm_eigenvalues<N>(h);
    asm volatile("" :: "v"(h.d[0]), "v"(h.d[3]));
    __builtin_amdgcn_sched_barrier(0);
    t[4] = stamp();
    double v[N];
    for (int a = 0; a < N; ++a) v[a] = vvd_from_sinh2(fmax(h.d[a], 0.0) * 0.25, 1e5);
    double acc = reduce_metric<N>(v, METRIC_RIEM, nullptr);
    if (!ok || !conv) acc = -1;
    if (i < b) out[i] = acc;
    __builtin_amdgcn_s_waitcnt(0x0070);
    t[5] = stamp();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* s = stamps + ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * NST;
        for (int k = 0; k < 6; ++k) s[k] = t[k];
    }
}

int main() {
    const long long B = 65536, NODES = 5041;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(0, 0.3);
    std::vector<double> tab(NODES * 32);
    for (long long r = 0; r < NODES; ++r) {
        double a[4][4], y[4][4];
        for (int i = 0; i < 4; ++i) for (int j = i; j < 4; ++j) { double v = nd(rng); tab[r * 32 + i * 4 + j] = v; tab[r * 32 + j * 4 + i] = v; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a[i][j] = nd(rng) + (i == j ? 1.0 : 0.0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { y[i][j] = 0; for (int k = 0; k < 4; ++k) y[i][j] += a[i][k] * a[j][k]; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) tab[r * 32 + 16 + i * 4 + j] = y[i][j] + (i == j ? 0.2 : 0.0);
    }
    std::vector<long long> idx(2 * B * 16);
    for (auto& v : idx) v = rng() % NODES;
    double *dt, *dout; long long* di; unsigned long long* ds;
    const int NW = 1024;
    CK(hipMalloc(&dt, tab.size() * 8)); CK(hipMalloc(&dout, B * 8)); CK(hipMalloc(&di, idx.size() * 8));
    CK(hipMalloc(&ds, NW * NST * 8));
    CK(hipMemcpy(dt, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, idx.data(), idx.size() * 8, hipMemcpyHostToDevice));
    const int LAUNCHES = 48, SKIP = 16;
    CK(hipFree(ds));
    CK(hipMalloc(&ds, (size_t)LAUNCHES * NW * NST * 8));
    const char* names[] = {"wave start", "idx loaded", "rows in registers", "front done (chol, solves, gram)", "eigenvalues done", "epilogue + store drained"};
    std::vector<double> mean(6, 0), mx(6, 0), mn(6, 1e30);
    for (int rep = 0; rep < 5; ++rep) {
        // back-to-back launches (steady clocks); each launch stamps into its own slice
        for (int r = 0; r < LAUNCHES; ++r) timeline<<<256, 256>>>(dt, di + (r % 16) * 2 * B, B, dout, ds + (size_t)r * NW * NST);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h((size_t)LAUNCHES * NW * NST);
    CK(hipMemcpy(h.data(), ds, h.size() * 8, hipMemcpyDeviceToHost));
    const int reps = LAUNCHES - SKIP;
    double launch_period = 0;
    unsigned long long prev_t0 = 0;
    for (int r = SKIP; r < LAUNCHES; ++r) {
        const unsigned long long* hh = h.data() + (size_t)r * NW * NST;
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < NW; ++w) t0 = std::min(t0, hh[w * NST]);
        if (r > SKIP) launch_period += (double)(t0 - prev_t0) * 0.01 / (reps - 1);
        prev_t0 = t0;
        for (int k = 0; k < 6; ++k) {
            double m = 0, x = 0, n0 = 1e30;
            for (int w = 0; w < NW; ++w) { double d = (double)(hh[w * NST + k] - t0) * 0.01; m += d; x = std::max(x, d); n0 = std::min(n0, d); }
            mean[k] += m / NW / reps; mx[k] = std::max(mx[k], x); mn[k] = std::min(mn[k], n0);
        }
    }
    printf("launch-to-launch period (first wave start to first wave start): %.2f us\n", launch_period);
    printf("stage boundary (us after the first wave of the launch started)      mean     min     max\n");
    for (int k = 0; k < 6; ++k) printf("  %-40s %7.2f %7.2f %7.2f\n", names[k], mean[k], mn[k], mx[k]);
    return 0;
}
