// Siegel backward, sixteen lanes per pair, upper-half model (siegel_coop_bwd_kernel.hpp): M = 9, dense output.
// One kernel per translation unit: the build checks each unit's ISA for the DPP copy hazard (tools/check_dpp_hazards.py)
// and only a unit that fails pays for the safe form.
#include "siegel_coop_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_coop_upper_9_dense(const BwdArgs& a, hipStream_t s) { return launch_coop_bwd_ms<9, false>(a, s); }

int launch_bwd_coop_upper(const BwdArgs& a, int n, bool scatter, hipStream_t s) {
    switch (n) {
        case 9: return scatter ? launch_bwd_coop_upper_9_scatter(a, s) : launch_bwd_coop_upper_9_dense(a, s);
        case 10: return scatter ? launch_bwd_coop_upper_10_scatter(a, s) : launch_bwd_coop_upper_10_dense(a, s);
        case 11: return scatter ? launch_bwd_coop_upper_11_scatter(a, s) : launch_bwd_coop_upper_11_dense(a, s);
        case 12: return scatter ? launch_bwd_coop_upper_12_scatter(a, s) : launch_bwd_coop_upper_12_dense(a, s);
        case 13: return scatter ? launch_bwd_coop_upper_13_scatter(a, s) : launch_bwd_coop_upper_13_dense(a, s);
        case 14: return scatter ? launch_bwd_coop_upper_14_scatter(a, s) : launch_bwd_coop_upper_14_dense(a, s);
        case 15: return scatter ? launch_bwd_coop_upper_15_scatter(a, s) : launch_bwd_coop_upper_15_dense(a, s);
        case 16: return scatter ? launch_bwd_coop_upper_16_scatter(a, s) : launch_bwd_coop_upper_16_dense(a, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "sixteen-lanes-per-pair backward covers dims 9..16");
    }
}
}  // namespace sympa_hip
