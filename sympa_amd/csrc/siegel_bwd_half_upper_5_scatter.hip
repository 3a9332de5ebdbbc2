// Siegel backward, EIGHT lanes per pair (two pairs per DPP row: SYMPA_COOP_HALF, spd_coop.hpp): upper model, M = 5,
// scatter output.  One kernel per translation unit (the build's DPP hazard check works per unit).
#define SYMPA_COOP_HALF
#define SYMPA_COOP_BWD_WAVES_UPPER 2      // two 256-register waves per SIMD: fused step n = 8 1663 -> 1487 us per 262 144 pairs
#include "siegel_coop_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_half_upper_5_scatter(const BwdArgs& a, hipStream_t s) { return launch_coop_bwd_ms<sympa::MODEL_UPPER, 5, true>(a, s); }
}  // namespace sympa_hip
