// Ablation of the n=4 upper kernel: time per launch (B = 65 536 pairs, 1 wave/SIMD)
// of cumulative stages, same arithmetic as the product (includes siegel_math.hpp).
//   stage 0: index + row loads, store            stage 1: + 2 Cholesky + D
//   stage 2: + 2 triangular solves               stage 3: + Gram matrix
//   stage 4: + K Jacobi sweeps (K = 1..5)        stage 5: + finishing sweep + epilogue (4 log1p) = full
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include "../../sympa_amd/csrc/siegel_math.hpp"

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
using namespace sympa;

template <int STAGE, int SWEEPS>
__global__ __launch_bounds__(256) void anat(const double* table, const long long* idx, long long b, double* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ii = i < b ? i : b - 1;
    const long long r1 = idx[2 * ii], r2 = idx[2 * ii + 1];
    constexpr int N = 4;
    CMat<N> z1, z2, e;
    load_point<N>(table + r1 * 32, z1);
    load_point<N>(table + r2 * 32, z2);
    double acc = 0.0;
    if (STAGE == 0) {
        for (int a = 0; a < N; ++a) for (int c = a; c < N; ++c) acc += z1.re[a][c] + z1.im[a][c] + z2.re[a][c] + z2.im[a][c];
    } else {
        Tri<N, false> l1, l2;
        bool ok = chol_real<N>(z1.im, l1);
        ok = chol_real<N>(z2.im, l2) && ok;
        for (int a = 0; a < N; ++a) for (int c = 0; c < N; ++c) { e.re[a][c] = z2.re[a][c] - z1.re[a][c]; e.im[a][c] = z2.im[a][c] - z1.im[a][c]; }
        if (STAGE == 1) {
            for (int a = 0; a < N; ++a) { acc += l1.rdiag[a] + l2.rdiag[a]; for (int c = 0; c < a; ++c) acc += l1.re[a][c] + l2.re[a][c]; }
            for (int a = 0; a < N; ++a) for (int c = a; c < N; ++c) acc += e.re[a][c] + e.im[a][c];
        } else {
            solve_left<N, false>(l1, e);
            solve_right_t<N, false>(l2, e);
            if (STAGE == 2) {
                for (int a = 0; a < N; ++a) for (int c = 0; c < N; ++c) acc += e.re[a][c] + e.im[a][c];
            } else {
                Herm<N> h;
                gram<N>(e, h);
                if (STAGE >= 4) {
#pragma unroll
                    for (int s = 0; s < SWEEPS; ++s) jacobi_sweep<N>(h);
                }
                if (STAGE == 5) {
                    jacobi_final_sweep<N>(h);
                    double v[N];
                    for (int a = 0; a < N; ++a) v[a] = vvd_from_sinh2(fmax(h.d[a], 0.0) * 0.25, 1e5);
                    acc = reduce_metric<N>(v, METRIC_RIEM, nullptr);
                } else {
                    for (int a = 0; a < N; ++a) { acc += h.d[a]; for (int c = a + 1; c < N; ++c) acc += h.re[a][c] + h.im[a][c]; }
                }
            }
        }
        if (!ok) acc = -1.0;
    }
    if (i < b) out[i] = acc;
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
    double *dt, *dout; long long* di;
    CK(hipMalloc(&dt, tab.size() * 8)); CK(hipMalloc(&dout, B * 8)); CK(hipMalloc(&di, idx.size() * 8));
    CK(hipMemcpy(dt, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, idx.data(), idx.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    const int reps = 320;
#define RUN(ST, SW, NAME) { \
        for (int r = 0; r < 32; ++r) anat<ST, SW><<<256, 256>>>(dt, di + (r % 16) * 2 * B, B, dout); \
        CK(hipEventRecord(s)); \
        for (int r = 0; r < reps; ++r) anat<ST, SW><<<256, 256>>>(dt, di + (r % 16) * 2 * B, B, dout); \
        CK(hipEventRecord(e)); CK(hipEventSynchronize(e)); float ms; CK(hipEventElapsedTime(&ms, s, e)); \
        double h0; CK(hipMemcpy(&h0, dout, 8, hipMemcpyDeviceToHost)); \
        printf("%-44s %7.3f us per launch   (out[0]=%g)\n", NAME, ms * 1e3 / reps, h0); }
    RUN(0, 0, "0 loads + store")
    RUN(1, 0, "1 + Cholesky x2 + D")
    RUN(2, 0, "2 + triangular solves")
    RUN(3, 0, "3 + Gram")
    RUN(4, 1, "4 + 1 Jacobi sweep")
    RUN(4, 2, "4 + 2 Jacobi sweeps")
    RUN(4, 3, "4 + 3 Jacobi sweeps")
    RUN(4, 4, "4 + 4 Jacobi sweeps")
    RUN(5, 3, "5 full: 3 sweeps + finish + epilogue")
    RUN(5, 4, "5 full: 4 sweeps + finish + epilogue")
    return 0;
}
