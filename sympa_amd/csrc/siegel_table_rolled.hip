// Optimiser-side table operations for dims 9..16 (C-ABI sympa_egrad2rgrad / sympa_projx / sympa_rsgd_step* /
// sympa_tangent_sqnorm): the SAME per-row arithmetic as dims <= 8 (siegel_table_math.hpp) compiled with rolled loops, the
// 16 x 16 working matrices in per-lane scratch.  Functional, not tuned: it completes the training path of the dims whose
// distances and gradients run sixteen lanes per pair.
#define SYMPA_UNROLL _Pragma("nounroll")
#include "siegel_table_kernel.hpp"

namespace sympa_hip {
int launch_table_rolled(int op, int n, int model, double* z, const double* g, double* out, int64_t b, double lr, double wd,
                        double eps, int32_t* projected, int32_t* status, hipStream_t s, const double* clip, double max_norm,
                        const int* gate) {
    switch (n) {
        case 9: return launch_table<9>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        case 10: return launch_table<10>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        case 11: return launch_table<11>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        case 12: return launch_table<12>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        case 13: return launch_table<13>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        case 14: return launch_table<14>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        case 15: return launch_table<15>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        case 16: return launch_table<16>(op, model, z, g, out, b, lr, wd, eps, projected, status, s, clip, max_norm, gate);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "rolled table operations cover dims 9..16");
    }
}
}  // namespace sympa_hip
