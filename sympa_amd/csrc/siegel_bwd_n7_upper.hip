// Backward kernels for n = 7, upper model (see siegel_bwd_kernel.hpp).
#include "siegel_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_n7_upper(const BwdArgs& a, bool scatter, hipStream_t s) { return launch_bwd_nm<7, sympa::MODEL_UPPER>(a, scatter, s); }
}  // namespace sympa_hip
