// Unit check of the eight-lanes-per-pair primitives (SYMPA_COOP_HALF): group_sum, Cholesky of eight 8 x 8 matrices per wave.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../sympa_amd/csrc -I../../include -o half_group_check half_group_check.hip
#define SYMPA_COOP_HALF
#include "spd_coop.hpp"
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* l, double* sums) {
    using namespace spd_coop;
    constexpr int M = 8;
    const int lane = threadIdx.x, g = lane / GROUP, r = lane % GROUP;
    double row[M], rd[M];
    for (int j = 0; j < M; ++j) row[j] = x[g * 64 + r * 8 + j];
    double s = 0.0;
    for (int j = 0; j < M; ++j) s += row[j];
    sums[lane] = group_sum(s);
    cholesky_rows(row, rd);
    for (int j = 0; j < M; ++j) l[g * 64 + r * 8 + j] = row[j];
}

int main() {
    std::vector<double> x(512), l(512), s(64);
    for (int g = 0; g < 8; ++g)
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) x[g * 64 + i * 8 + j] = (i == j ? 8.0 + g : 0.0) + 0.1 * ((i * 7 + j * 3 + g) % 5 + (j * 7 + i * 3 + g) % 5);
    double *dx, *dl, *ds;
    (void)hipMalloc(&dx, 512 * 8); (void)hipMalloc(&dl, 512 * 8); (void)hipMalloc(&ds, 64 * 8);
    (void)hipMemcpy(dx, x.data(), 512 * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dl, ds);
    (void)hipMemcpy(l.data(), dl, 512 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(s.data(), ds, 64 * 8, hipMemcpyDeviceToHost);
    double worst = 0, worst_s = 0;
    for (int g = 0; g < 8; ++g) {
        double tot = 0; for (int i = 0; i < 64; ++i) tot += x[g * 64 + i];
        for (int r = 0; r < 8; ++r) worst_s = fmax(worst_s, fabs(s[g * 8 + r] - tot));
        // L L^T == X (lower triangle of L is what the routine leaves in registers j <= i)
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j <= i; ++j) {
                double t = 0; for (int kk = 0; kk <= j; ++kk) t += l[g * 64 + i * 8 + kk] * l[g * 64 + j * 8 + kk];
                worst = fmax(worst, fabs(t - x[g * 64 + i * 8 + j]));
            }
    }
    printf("group_sum error %.3e   Cholesky reconstruction error %.3e\n", worst_s, worst);
    // reference factor of group 1 on the host
    {
        const int g = 1; double L[8][8] = {};
        for (int j = 0; j < 8; ++j) {
            double t = x[g * 64 + j * 8 + j]; for (int k = 0; k < j; ++k) t -= L[j][k] * L[j][k];
            L[j][j] = sqrt(t);
            for (int i = j + 1; i < 8; ++i) { double u = x[g * 64 + i * 8 + j]; for (int k = 0; k < j; ++k) u -= L[i][k] * L[j][k]; L[i][j] = u / L[j][j]; }
        }
        for (int i = 0; i < 8; ++i) { for (int j = 0; j <= i; ++j) printf(" %9.2e", l[g * 64 + i * 8 + j] - L[i][j]); printf("\n"); }
    }
    return 0;
}
