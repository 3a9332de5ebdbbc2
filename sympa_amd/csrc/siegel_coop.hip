// Dispatch of the upper and bounded models for 9 <= n <= 16 (sixteen lanes per pair, siegel_coop_kernel.hpp).
#include "siegel_coop_kernel.hpp"

namespace sympa_hip {

int launch_siegel_coop(const DistArgs& a, int n, int model, hipStream_t s) {
    return model == SYMPA_MODEL_UPPER ? launch_siegel_coop_upper(a, n, s) : launch_siegel_coop_bounded(a, n, s);
}

}  // namespace sympa_hip
