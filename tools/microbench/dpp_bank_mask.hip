// Does the DP-ALU DPP form (v_fmac_f64_dpp row_newbcast) honour bank_mask?  And the 64-bit update_dpp the compiler emits?
// hipcc --offload-arch=gfx950 -O2 -o dpp_bank_mask dpp_bank_mask.hip && ./dpp_bank_mask
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(double* out) {
    const int lane = threadIdx.x;
    double x = 100.0 + lane, y = 1.0, acc = 0.0, acc2 = 0.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(acc) : "v"(x), "v"(y));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(x), "v"(y));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(x), "v"(y));
    const long long lo = __builtin_amdgcn_update_dpp(0ll, __double_as_longlong(x), 0x150 + 3, 0xf, 0x3, true);
    const double b = __longlong_as_double(__builtin_amdgcn_update_dpp(lo, __double_as_longlong(x), 0x158 + 3, 0xf, 0xc, true));
    double acc3 = 5.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(acc3) : "v"(x), "v"(y));
    const double mid = acc3;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(acc3) : "v"(x), "v"(y));
    double acc4 = 5.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(acc4) : "v"(x), "v"(y));
    double acc5 = 5.0, acc6 = 5.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(acc5) : "v"(x), "v"(y));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc6) : "v"(x), "v"(y));
    out[384 + lane] = acc5; out[448 + lane] = acc6;
    out[lane] = acc; out[64 + lane] = acc2; out[128 + lane] = b; out[192 + lane] = mid; out[256 + lane] = acc3; out[320 + lane] = acc4;
}

int main() {
    double* d; hipMalloc(&d, 512 * 8);
    k<<<1, 64>>>(d);
    double h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("fmac with bank masks (want 103 x8, 111 x8):"); for (int i = 0; i < 16; ++i) printf(" %g", h[i]); printf("\n");
    printf("fmac full mask (want 103 x16):             "); for (int i = 0; i < 16; ++i) printf(" %g", h[64 + i]); printf("\n");
    printf("update_dpp pair (want 103 x8, 111 x8):     "); for (int i = 0; i < 16; ++i) printf(" %g", h[128 + i]); printf("\n");
    printf("acc = 5 after the first masked fmac (want 108 x8, 5 x8):  "); for (int i = 0; i < 16; ++i) printf(" %g", h[192 + i]); printf("\n");
    printf("after both (want 108 x8, 116 x8):                       "); for (int i = 0; i < 16; ++i) printf(" %g", h[256 + i]); printf("\n");
    printf("negated, both in one statement (want -98 x8, -106 x8):   "); for (int i = 0; i < 16; ++i) printf(" %g", h[320 + i]); printf("\n");
    printf("negated, bank_mask 0x3 alone (want -98 x8, 5 x8):        "); for (int i = 0; i < 16; ++i) printf(" %g", h[384 + i]); printf("\n");
    printf("negated, full mask (want -98 x16):                       "); for (int i = 0; i < 16; ++i) printf(" %g", h[448 + i]); printf("\n");
    return 0;
}
