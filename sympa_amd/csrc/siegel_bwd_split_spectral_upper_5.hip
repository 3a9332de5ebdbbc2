// Split Siegel backward, stage 1 (eigen-decomposition with vectors -> Hbar, K): upper model, n = 5, one pair per lane.
// One kernel per translation unit (the unrolled kernels compile in parallel).  See siegel_bwd_split_kernel.hpp.
#include "siegel_bwd_split_kernel.hpp"

namespace sympa_hip {
int launch_bwd_split_spectral_upper_5(const SplitArgs& sa, hipStream_t s) { return launch_bwd_split_spectral<5, sympa::MODEL_UPPER>(sa, s); }
}  // namespace sympa_hip
