// Siegel backward, sixteen lanes per pair (siegel_coop_bwd_kernel.hpp): upper model, M = 7, dense output -- for the A/B
// against the one-pair-per-lane kernel only (SYMPA_FLAG_COOP; half the lanes of a group are phantoms at this size).
#include "siegel_coop_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_coop_upper_7_dense(const BwdArgs& a, hipStream_t s) { return launch_coop_bwd_ms<sympa::MODEL_UPPER, 7, false>(a, s); }
}  // namespace sympa_hip
