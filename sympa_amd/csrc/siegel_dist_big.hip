// Forward, one pair per lane, dims 5..8 (siegel_dist_kernel.hpp).  This unit is compiled with -mllvm -enable-misched=0
// (see the header and __graft_entry__.py).
#include "siegel_dist_kernel.hpp"

namespace sympa_hip {

int launch_dist_big(const DistArgs& a, int n, int model, hipStream_t s) {
    switch (n) {
        case 5: return launch_n<5>(a, model, s);
        case 6: return launch_n<6>(a, model, s);
        case 7: return launch_n<7>(a, model, s);
        case 8: return launch_n<8>(a, model, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "dims 5..8");
    }
}

int launch_multi_big(const MultiArgs& m, unsigned grid, int n, int model, hipStream_t s) {
    switch (n) {
        case 5: return launch_multi_n<5>(m, grid, model, s);
        case 6: return launch_multi_n<6>(m, grid, model, s);
        case 7: return launch_multi_n<7>(m, grid, model, s);
        case 8: return launch_multi_n<8>(m, grid, model, s);
        default: return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "dims 5..8");
    }
}

}  // namespace sympa_hip
