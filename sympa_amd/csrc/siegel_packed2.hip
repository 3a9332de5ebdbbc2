// The two-waves-per-SIMD packed forward of the upper model at dims 7, 8 (siegel_packed2_kernel.hpp): dispatch for siegel_packed.hip.
#include <cstdlib>
#include "siegel_packed2_kernel.hpp"

namespace sympa_hip {

int launch_packed_forward2_n(const PackedArgs& a0, int n, unsigned cus, hipStream_t s) {
    PackedArgs a = a0;
    const char* e = getenv("SYMPA_P2_STAGGER");
    a.stagger = e ? atoi(e) : 0;
    switch (n) {
        case 7: return launch_packed_forward2<7>(a, cus, s);
        case 8: return launch_packed_forward2<8>(a, cus, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "two-waves packed forward: dims 7, 8");
}

}  // namespace sympa_hip
