// Split Siegel backward, stage 2 (factors and E again, products / solves / congruences from Hbar, K): upper model, n = 6,
// atomic scatter into the table gradient.  One kernel per translation unit.  See siegel_bwd_split_kernel.hpp.
#include "siegel_bwd_split_kernel.hpp"

namespace sympa_hip {
int launch_bwd_split_gradient_upper_6_scatter(const SplitArgs& sa, hipStream_t s) { return launch_bwd_split_gradient<6, sympa::MODEL_UPPER, true>(sa, s); }
}  // namespace sympa_hip
