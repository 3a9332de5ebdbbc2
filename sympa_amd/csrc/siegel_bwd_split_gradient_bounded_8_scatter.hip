// Split Siegel backward, stage 2 (factors and E again, products / solves / congruences from Hbar, K): bounded model, n = 8,
// atomic scatter into the table gradient.  One kernel per translation unit.  See siegel_bwd_split_kernel.hpp.
#include "siegel_bwd_split_kernel.hpp"

namespace sympa_hip {
int launch_bwd_split_gradient_bounded_8_scatter(const SplitArgs& sa, hipStream_t s) { return launch_bwd_split_gradient<8, sympa::MODEL_BOUNDED, true>(sa, s); }
}  // namespace sympa_hip
