// C-ABI sympa_table_digest + the launcher the pack units share (table_digest.hpp).
#include "siegel_common.hpp"
#include "table_digest.hpp"

namespace sympa_hip {

// word i of the table enters as mix(w) * (odd multiplier of i): a change of any single word changes the sum (odd multipliers are
// units mod 2^64); sums commute, so blocks accumulate with one 64-bit atomic each, in any order, to the same value
__device__ __forceinline__ unsigned long long digest_term(unsigned long long w, unsigned long long i) {
    w ^= w >> 29;
    return w * ((i * 0x9E3779B97F4A7C15ull) | 1ull);
}

__global__ __launch_bounds__(DIGEST_BLOCK) void table_digest_kernel(const unsigned long long* __restrict__ data, const int64_t words,
                                                                    unsigned long long* __restrict__ state, const int force) {
    __shared__ unsigned long long part[DIGEST_BLOCK / 64];
    unsigned long long acc = 0;
    const int64_t pairs = words >> 1;
    const int64_t stride = (int64_t)gridDim.x * DIGEST_BLOCK;
    const ulonglong2* d2 = reinterpret_cast<const ulonglong2*>(data);
    for (int64_t p = (int64_t)blockIdx.x * DIGEST_BLOCK + threadIdx.x; p < pairs; p += stride) {
        const ulonglong2 v = d2[p];
        acc += digest_term(v.x, (unsigned long long)(2 * p));
        acc += digest_term(v.y, (unsigned long long)(2 * p + 1));
    }
    if ((words & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += digest_term(data[words - 1], (unsigned long long)(words - 1));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
#pragma unroll
        for (int k = 0; k < DIGEST_BLOCK / 64; ++k) s += part[k];
        atomicAdd(&state[1], s);
        __threadfence();
        unsigned* u = reinterpret_cast<unsigned*>(state);
        const unsigned ticket = atomicAdd(&u[4], 1u);
        if (ticket == gridDim.x - 1) {                     // last block: every partial sum is in
            __threadfence();
            const unsigned long long total = atomicAdd(&state[1], 0ull);
            const unsigned changed = (force || total != state[0]) ? 1u : 0u;
            state[0] = total;
            state[1] = 0ull;
            u[4] = 0u;
            u[DIGEST_GUARD_WORD] = changed;
            u[DIGEST_GUARD_WORD + 1] += changed;
        }
    }
}

int launch_table_digest(const void* data, int64_t bytes, void* state, int force, hipStream_t s) {
    if (data == nullptr || state == nullptr) return fail(SYMPA_ERR_BAD_ARG, "table digest: null buffer");
    if (bytes <= 0 || (bytes & 7)) return fail(SYMPA_ERR_BAD_ARG, "table digest: a positive multiple of 8 bytes");
    if ((reinterpret_cast<uintptr_t>(data) & 15) || (reinterpret_cast<uintptr_t>(state) & 7))
        return fail(SYMPA_ERR_BAD_ARG, "table digest: data 16-byte aligned, state 8-byte aligned");
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, c = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)
            c = 256;
        cus = c;
    }
    const int64_t words = bytes >> 3;
    // 32 bytes per thread and trip: enough blocks to cover the table once, at most eight per CU (the loop strides beyond that)
    int64_t blocks = ((words >> 1) + DIGEST_BLOCK - 1) / DIGEST_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > (int64_t)cus * 8) blocks = (int64_t)cus * 8;
    hipLaunchKernelGGL(table_digest_kernel, dim3((unsigned)blocks), dim3(DIGEST_BLOCK), 0, s,
                       reinterpret_cast<const unsigned long long*>(data), words, reinterpret_cast<unsigned long long*>(state), force);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace sympa_hip

extern "C" {

int sympa_table_digest(const void* data, int64_t bytes, void* state, int flags, void* stream) {
    return sympa_hip::launch_table_digest(data, bytes, state, (flags & SYMPA_FLAG_DIGEST_FORCE) ? 1 : 0,
                                          reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
