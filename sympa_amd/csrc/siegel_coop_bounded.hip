// Sixteen-lanes-per-pair kernels of the bounded model, dims 9..16 (siegel_coop_kernel.hpp).
#include "siegel_coop_kernel.hpp"

namespace sympa_hip {
int launch_siegel_coop_bounded(const DistArgs& a, int n, hipStream_t s) {
    return launch_siegel_coop_model<sympa::MODEL_BOUNDED>(a, n, s);
}
}  // namespace sympa_hip
