// Backward kernels for n = 8, bounded model (see siegel_bwd_kernel.hpp).
#include "siegel_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_n8_bounded(const BwdArgs& a, bool scatter, hipStream_t s) { return launch_bwd_nm<8, sympa::MODEL_BOUNDED>(a, scatter, s); }
}  // namespace sympa_hip
