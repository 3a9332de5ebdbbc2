// Siegel backward, sixteen lanes per pair: dispatch over model, size and output form (kernels: siegel_bwd_coop_*_*.hip)
#include "siegel_coop_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_coop(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s) {
    const bool upper = model == SYMPA_MODEL_UPPER;
    switch (n) {
        case 9:
            if (upper) return scatter ? launch_bwd_coop_upper_9_scatter(a, s) : launch_bwd_coop_upper_9_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_9_scatter(a, s) : launch_bwd_coop_bounded_9_dense(a, s);
        case 10:
            if (upper) return scatter ? launch_bwd_coop_upper_10_scatter(a, s) : launch_bwd_coop_upper_10_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_10_scatter(a, s) : launch_bwd_coop_bounded_10_dense(a, s);
        case 11:
            if (upper) return scatter ? launch_bwd_coop_upper_11_scatter(a, s) : launch_bwd_coop_upper_11_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_11_scatter(a, s) : launch_bwd_coop_bounded_11_dense(a, s);
        case 12:
            if (upper) return scatter ? launch_bwd_coop_upper_12_scatter(a, s) : launch_bwd_coop_upper_12_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_12_scatter(a, s) : launch_bwd_coop_bounded_12_dense(a, s);
        case 13:
            if (upper) return scatter ? launch_bwd_coop_upper_13_scatter(a, s) : launch_bwd_coop_upper_13_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_13_scatter(a, s) : launch_bwd_coop_bounded_13_dense(a, s);
        case 14:
            if (upper) return scatter ? launch_bwd_coop_upper_14_scatter(a, s) : launch_bwd_coop_upper_14_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_14_scatter(a, s) : launch_bwd_coop_bounded_14_dense(a, s);
        case 15:
            if (upper) return scatter ? launch_bwd_coop_upper_15_scatter(a, s) : launch_bwd_coop_upper_15_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_15_scatter(a, s) : launch_bwd_coop_bounded_15_dense(a, s);
        case 16:
            if (upper) return scatter ? launch_bwd_coop_upper_16_scatter(a, s) : launch_bwd_coop_upper_16_dense(a, s);
            return scatter ? launch_bwd_coop_bounded_16_scatter(a, s) : launch_bwd_coop_bounded_16_dense(a, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "sixteen-lanes-per-pair backward covers dims 9..16");
    }
}
}  // namespace sympa_hip
