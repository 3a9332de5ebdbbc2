// TEST INFRASTRUCTURE ONLY (CPU).  Compiles sympa_amd/csrc/siegel_math.hpp with g++ so the exact
// per-pair arithmetic the gfx950 kernels run can be checked against the oracle and the golden
// vectors in a container without a GPU.  Nothing in sympa_amd/ loads this library; the product
// path fails loudly when the HIP library is missing.
#include <cstdint>
#include "../../sympa_amd/csrc/siegel_math.hpp"

namespace {
template <int N>
void run(const double* z1, const double* z2, int64_t b, int model, int metric, const double* w, double eps,
         double* out, double* vvd, int32_t* status) {
    int st = 0;
    for (int64_t i = 0; i < b; ++i) {
        const double* p1 = z1 + i * 2 * N * N;
        const double* p2 = z2 + i * 2 * N * N;
        double* vv = vvd ? vvd + i * N : nullptr;
        out[i] = (model == sympa::MODEL_UPPER)
                     ? sympa::pair_distance<N, sympa::MODEL_UPPER>(p1, p2, metric, w, 1.0 / eps, vv, st)
                     : sympa::pair_distance<N, sympa::MODEL_BOUNDED>(p1, p2, metric, w, 1.0 / eps, vv, st);
    }
    if (status) *status = st;
}
}  // namespace

extern "C" int sympa_hostsim_dist(const double* z1, const double* z2, int64_t b, int n, int model, int metric,
                                  const double* w, double eps, double* out, double* vvd, int32_t* status) {
    switch (n) {
        case 1: run<1>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 2: run<2>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 3: run<3>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 4: run<4>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 5: run<5>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 6: run<6>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 7: run<7>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 8: run<8>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        default: return -2;
    }
}
