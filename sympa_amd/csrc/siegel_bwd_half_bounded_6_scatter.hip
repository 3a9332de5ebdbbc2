// Siegel backward, EIGHT lanes per pair (two pairs per DPP row: SYMPA_COOP_HALF, spd_coop.hpp): bounded model, M = 6,
// scatter output.  One kernel per translation unit (the build's DPP hazard check works per unit).
#define SYMPA_COOP_HALF
#include "siegel_coop_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_half_bounded_6_scatter(const BwdArgs& a, hipStream_t s) { return launch_coop_bwd_ms<sympa::MODEL_BOUNDED, 6, true>(a, s); }
}  // namespace sympa_hip
