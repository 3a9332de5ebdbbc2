// Split Siegel backward, stage 2 (factors and E again, products / solves / congruences from Hbar, K): bounded model, n = 6,
// per-pair gradient rows.  One kernel per translation unit.  See siegel_bwd_split_kernel.hpp.
#include "siegel_bwd_split_kernel.hpp"

namespace sympa_hip {
int launch_bwd_split_gradient_bounded_6_dense(const SplitArgs& sa, hipStream_t s) { return launch_bwd_split_gradient<6, sympa::MODEL_BOUNDED, false>(sa, s); }
}  // namespace sympa_hip
