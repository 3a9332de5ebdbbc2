#define SYMPA_COOP_HALF
#include "spd_coop.hpp"
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o) {
    using namespace spd_coop;
    const int lane = threadIdx.x, g = lane / GROUP, r = lane % GROUP;
    double x0 = x[g * 64 + r * 8 + 0], x1 = x[g * 64 + r * 8 + 1];
    const double piv = bcast<0>(settle(x0));
    const double rr = 1.0 / sqrt(piv);
    x0 = settle(x0 * rr);
    o[lane] = x0;                 // L[i][0]
    const double before = x1;
    fnmac_bc<1>(x1, x0, x0);
    o[64 + lane] = x1;            // X[i][1] - L[1][0] L[i][0]
    o[128 + lane] = bcast<1>(settle(x1));
    o[192 + lane] = before;
    o[256 + lane] = bcast<1>(x0);
}
int main() {
    std::vector<double> x(512), o(320);
    for (int g = 0; g < 8; ++g) for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j)
        x[g * 64 + i * 8 + j] = (i == j ? 8.0 + g : 0.0) + 0.1 * ((i * 7 + j * 3 + g) % 5 + (j * 7 + i * 3 + g) % 5);
    double *dx, *dо; (void)hipMalloc(&dx, 512 * 8); (void)hipMalloc(&dо, 320 * 8);
    (void)hipMemcpy(dx, x.data(), 512 * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dо);
    (void)hipMemcpy(o.data(), dо, 320 * 8, hipMemcpyDeviceToHost);
    for (int lane = 8; lane < 16; ++lane)
        printf("lane %2d: L[i][0] %.6f  x1 before %.6f  after %.6f (want %.6f)  bcast<1>(x1) %.6f  bcast<1>(x0) %.6f\n", lane, o[lane], o[192 + lane], o[64 + lane],
               o[192 + lane] - o[8 + 1] * o[lane], o[128 + lane], o[256 + lane]);
    return 0;
}
