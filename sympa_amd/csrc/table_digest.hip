// C-ABI sympa_table_digest + the launcher the pack units share (table_digest.hpp).
#include "siegel_common.hpp"
#include "table_digest.hpp"

namespace sympa_hip {

// 16-byte chunk p of the table enters as mix(x) * m(p) + mix(y) * m'(p) with odd 64-bit multipliers made from p: a change of any
// single word changes the sum (odd multipliers are units mod 2^64); sums commute, so the blocks' partial sums can be added in any
// order to the same value
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long digest_chunk(const u64x2 v, const unsigned long long p) {
    const unsigned long long m = (p * 0x9E3779B97F4A7C15ull) | 1ull;
    unsigned long long x = v.x, y = v.y;
    x ^= x >> 29;
    y ^= y >> 31;
    return x * m + y * ((m ^ 0xD6E8FEB86659FD93ull) | 1ull);
}

// Round 6, first form: 2 048 blocks, each adding its sum to ONE accumulator with a 64-bit atomic and taking a ticket with a
// second one -- 59 us for 46.6 MB (0.8 TB/s): 4 096 same-address atomics serialise at ~15 ns each.  Now: at most one block per CU,
// partial sums stored plainly, ONE atomic per block (the ticket); the last block adds the <= 256 partial sums.
__global__ __launch_bounds__(DIGEST_BLOCK) void table_digest_kernel(const unsigned long long* __restrict__ data, const int64_t words,
                                                                    unsigned long long* __restrict__ state,
                                                                    unsigned long long* __restrict__ partial, const int force) {
    __shared__ unsigned long long part[DIGEST_BLOCK / 64];
    __shared__ unsigned last;
    unsigned long long acc = 0;
    const int64_t chunks = words >> 1;
    const int64_t stride = (int64_t)gridDim.x * DIGEST_BLOCK;
    const u64x2* d2 = reinterpret_cast<const u64x2*>(data);
    int64_t p = (int64_t)blockIdx.x * DIGEST_BLOCK + threadIdx.x;
    // (plain loads: the forward that follows reads the same rows out of the caches)
    // eight independent 16-byte loads in flight per thread (256 blocks x 1 024 threads x 128 B = 32 MB per trip: configs[3]'s table
    // in two trips; with four the digest of its 46.6 MB took 16.5 us = 2.8 TB/s)
    for (; p + 7 * stride < chunks; p += 8 * stride) {
        u64x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = d2[p + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += digest_chunk(v[k], (unsigned long long)(p + k * stride));
    }
    for (; p < chunks; p += stride) acc += digest_chunk(d2[p], (unsigned long long)p);
    if ((words & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        u64x2 v;
        v.x = data[words - 1];
        v.y = 0ull;
        acc += digest_chunk(v, (unsigned long long)chunks);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    unsigned* u = reinterpret_cast<unsigned*>(state);
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
#pragma unroll
        for (int k = 0; k < DIGEST_BLOCK / 64; ++k) s += part[k];
        __hip_atomic_store(&partial[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned ticket = atomicAdd(&u[4], 1u);
        last = (ticket == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last == 0u) return;
    // last block: every partial sum is in (each was stored before its block's ticket)
    __threadfence();
    unsigned long long t = 0;
    for (unsigned k = threadIdx.x; k < gridDim.x; k += DIGEST_BLOCK)
        t += __hip_atomic_load(&partial[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = 0;
#pragma unroll
        for (int k = 0; k < DIGEST_BLOCK / 64; ++k) total += part[k];
        const unsigned changed = (force || total != state[0]) ? 1u : 0u;
        state[0] = total;
        u[4] = 0u;
        u[DIGEST_GUARD_WORD] = changed;
        u[DIGEST_GUARD_WORD + 1] += changed;
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
    // one 1 024-thread block per CU at most (DIGEST_MAX_BLOCKS partial sums behind the 32 bytes of state words)
    int64_t blocks = ((words >> 1) + 4 * DIGEST_BLOCK - 1) / (4 * DIGEST_BLOCK);
    if (blocks < 1) blocks = 1;
    if (blocks > cus) blocks = cus;
    if (blocks > DIGEST_MAX_BLOCKS) blocks = DIGEST_MAX_BLOCKS;
    unsigned long long* st = reinterpret_cast<unsigned long long*>(state);
    hipLaunchKernelGGL(table_digest_kernel, dim3((unsigned)blocks), dim3(DIGEST_BLOCK), 0, s,
                       reinterpret_cast<const unsigned long long*>(data), words, st, st + 4, force);
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
