// Forward, dims 7 and 8 with EIGHT lanes per pair (two pairs per DPP row: SYMPA_COOP_HALF, spd_coop.hpp), both models.
// A/B only (SYMPA_FLAG_COOP): measured against the one-pair-per-lane register kernels in profiles/r03_n8_forward_ab.txt.
#define SYMPA_COOP_HALF
#include "siegel_coop_kernel.hpp"

namespace sympa_hip {
int launch_siegel_coop_half(const DistArgs& a, int n, int model, hipStream_t s) {
    const bool up = model == SYMPA_MODEL_UPPER;
    switch (n) {
        case 7: return up ? launch_siegel_coop_m<sympa::MODEL_UPPER, 7>(a, s) : launch_siegel_coop_m<sympa::MODEL_BOUNDED, 7>(a, s);
        case 8: return up ? launch_siegel_coop_m<sympa::MODEL_UPPER, 8>(a, s) : launch_siegel_coop_m<sympa::MODEL_BOUNDED, 8>(a, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "eight-lanes-per-pair forward: dims 7, 8");
    }
}
}  // namespace sympa_hip
