// Backward kernel for n = 7, upper model, scatter into the table gradient (see siegel_bwd_kernel.hpp).
#include "siegel_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_n7_upper_scatter(const BwdArgs& a, hipStream_t s) { return launch_bwd_nms<7, sympa::MODEL_UPPER, true>(a, s); }
}  // namespace sympa_hip
