// Backward kernels for dims 9..16 (C-ABI sympa_siegel_dist_bwd / sympa_model_backward / sympa_model_loss_backward*):
// the SAME per-pair adjoint as dims <= 8 (siegel_math_bwd.hpp), compiled with rolled loops -- SYMPA_UNROLL = nounroll --
// so that the 16 x 16 working matrices are per-lane scratch arrays instead of registers.  One pair per lane, 64-thread
// blocks.  Functional, not tuned: it completes the drop-in (the reference differentiates every dims it accepts,
// runner.py:105); the forward of these dims runs sixteen lanes per pair (siegel_coop_kernel.hpp).
#define SYMPA_UNROLL _Pragma("nounroll")
#include "siegel_bwd_kernel.hpp"

namespace sympa_hip {

template <int N>
int launch_bwd_rolled_n(const BwdArgs& a, int model, bool scatter, hipStream_t s) {
    return model == SYMPA_MODEL_UPPER ? launch_bwd_nm<N, sympa::MODEL_UPPER>(a, scatter, s)
                                      : launch_bwd_nm<N, sympa::MODEL_BOUNDED>(a, scatter, s);
}

int launch_bwd_rolled(const BwdArgs& a, int n, int model, bool scatter, hipStream_t s) {
    switch (n) {
        case 9: return launch_bwd_rolled_n<9>(a, model, scatter, s);
        case 10: return launch_bwd_rolled_n<10>(a, model, scatter, s);
        case 11: return launch_bwd_rolled_n<11>(a, model, scatter, s);
        case 12: return launch_bwd_rolled_n<12>(a, model, scatter, s);
        case 13: return launch_bwd_rolled_n<13>(a, model, scatter, s);
        case 14: return launch_bwd_rolled_n<14>(a, model, scatter, s);
        case 15: return launch_bwd_rolled_n<15>(a, model, scatter, s);
        case 16: return launch_bwd_rolled_n<16>(a, model, scatter, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "rolled backward covers dims 9..16");
    }
}

}  // namespace sympa_hip
