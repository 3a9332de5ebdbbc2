// Backward kernels for n = 8, upper model (see siegel_bwd_kernel.hpp).
#include "siegel_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_n8_upper(const BwdArgs& a, bool scatter, hipStream_t s) { return launch_bwd_nm<8, sympa::MODEL_UPPER>(a, scatter, s); }
}  // namespace sympa_hip
