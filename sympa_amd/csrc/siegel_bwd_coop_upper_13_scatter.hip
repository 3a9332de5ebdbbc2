// Siegel backward, sixteen lanes per pair (siegel_coop_bwd_kernel.hpp): upper model, M = 13, scatter output.
// One kernel per translation unit: the build checks each unit's ISA for the DPP copy hazard (tools/check_dpp_hazards.py)
// and only a unit that fails pays for the safe form.
#include "siegel_coop_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_coop_upper_13_scatter(const BwdArgs& a, hipStream_t s) { return launch_coop_bwd_ms<sympa::MODEL_UPPER, 13, true>(a, s); }
}  // namespace sympa_hip
