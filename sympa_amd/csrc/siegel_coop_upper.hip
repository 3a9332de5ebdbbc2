// Sixteen-lanes-per-pair kernels of the upper model, dims 9..16 (siegel_coop_kernel.hpp).
#include "siegel_coop_kernel.hpp"

namespace sympa_hip {
int launch_siegel_coop_upper(const DistArgs& a, int n, hipStream_t s) {
    return launch_siegel_coop_model<sympa::MODEL_UPPER>(a, n, s);
}
}  // namespace sympa_hip
