// Backward kernel for n = 8, bounded model, scatter into the table gradient (see siegel_bwd_kernel.hpp).
#include "siegel_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_n8_bounded_scatter(const BwdArgs& a, hipStream_t s) { return launch_bwd_nms<8, sympa::MODEL_BOUNDED, true>(a, s); }
}  // namespace sympa_hip
