// C-ABI sympa_clock_stamp: the measured shader clock of a stream-ordered region (bench.py, round 6).
// One wave per block, SYMPA_CLOCK_STAMP_BLOCKS blocks (several per CU): every block writes {s_memtime (shader cycles), s_memrealtime
// (the constant 100 MHz counter), where it ran: XCC_ID | HW_ID << 8}.  The cycle counters of different CUs are NOT comparable
// (measured: pairing stamps by XCD alone gives offsets of ~1e7 cycles), so the host pairs the two stamps of the SAME CU (XCC, SE,
// SH, CU fields): clock = d(s_memtime) / d(s_memrealtime) x 100 MHz over exactly that region (MI355X_MICROARCH.md, "DVFS give-back"
// item 6) -- the real kernels carry no stamps.
#include "siegel_common.hpp"

namespace {

__global__ __launch_bounds__(64) void clock_stamp_kernel(unsigned long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    unsigned long long t, r;
    unsigned xcc, hw;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) :: "memory");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[3 * blockIdx.x + 0] = t;
    out[3 * blockIdx.x + 1] = r;
    out[3 * blockIdx.x + 2] = (unsigned long long)(xcc & 0xfu) | ((unsigned long long)hw << 8);
}

}  // namespace

extern "C" {

int sympa_clock_stamp(void* out, void* stream) {
    using namespace sympa_hip;
    if (out == nullptr || (reinterpret_cast<uintptr_t>(out) & 7)) return fail(SYMPA_ERR_BAD_ARG, "clock stamp: an 8-byte aligned device buffer");
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(SYMPA_CLOCK_STAMP_BLOCKS), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<unsigned long long*>(out));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
