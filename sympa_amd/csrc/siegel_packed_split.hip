// The two-kernel packed forward of the upper model at dims 7, 8 (siegel_packed_split.hpp): dispatch for siegel_packed.hip.
#include "siegel_packed_split.hpp"

namespace sympa_hip {

int64_t packed_split_workspace_bytes(int64_t b, int n, int model) {
    if (b <= 0 || model != SYMPA_MODEL_UPPER || !packed_split_dims_ok(n)) return 0;
    return ((b + 63) / 64) * (int64_t)(n * n + 1) * 64 * 8;
}

int launch_packed_split_n(const PackedArgs& a, int n, hipStream_t s) {
    switch (n) {
        case 7: return launch_packed_split<7>(a, s);
        case 8: return launch_packed_split<8>(a, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "split packed forward: dims 7, 8");
}

}  // namespace sympa_hip
