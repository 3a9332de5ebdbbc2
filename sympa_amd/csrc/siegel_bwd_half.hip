// Siegel backward, eight lanes per pair (SYMPA_COOP_HALF kernels, siegel_bwd_half_*_*.hip): dispatch, dims 5..8
#include "siegel_bwd_kernel.hpp"

namespace sympa_hip {
int launch_bwd_half_upper_5_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_upper_5_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_upper_6_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_upper_6_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_upper_7_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_upper_7_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_upper_8_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_upper_8_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_5_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_5_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_6_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_6_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_7_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_7_scatter(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_8_dense(const BwdArgs& a, hipStream_t s);
int launch_bwd_half_bounded_8_scatter(const BwdArgs& a, hipStream_t s);

int launch_bwd_half(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s) {
    const bool upper = model == SYMPA_MODEL_UPPER;
    switch (n) {
        case 5:
            if (upper) return scatter ? launch_bwd_half_upper_5_scatter(a, s) : launch_bwd_half_upper_5_dense(a, s);
            return scatter ? launch_bwd_half_bounded_5_scatter(a, s) : launch_bwd_half_bounded_5_dense(a, s);
        case 6:
            if (upper) return scatter ? launch_bwd_half_upper_6_scatter(a, s) : launch_bwd_half_upper_6_dense(a, s);
            return scatter ? launch_bwd_half_bounded_6_scatter(a, s) : launch_bwd_half_bounded_6_dense(a, s);
        case 7:
            if (upper) return scatter ? launch_bwd_half_upper_7_scatter(a, s) : launch_bwd_half_upper_7_dense(a, s);
            return scatter ? launch_bwd_half_bounded_7_scatter(a, s) : launch_bwd_half_bounded_7_dense(a, s);
        case 8:
            if (upper) return scatter ? launch_bwd_half_upper_8_scatter(a, s) : launch_bwd_half_upper_8_dense(a, s);
            return scatter ? launch_bwd_half_bounded_8_scatter(a, s) : launch_bwd_half_bounded_8_dense(a, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "eight-lanes-per-pair backward covers dims 5..8");
    }
}
}  // namespace sympa_hip
