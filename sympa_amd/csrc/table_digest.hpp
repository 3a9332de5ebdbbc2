// Device-side validity of a packed table (C-ABI sympa_table_digest, sympa_table_pack_refresh, sympa_spd_table_pack_refresh).
//
// A pack (sympa_table_pack / sympa_spd_table_pack) is an image of the embedding table at ONE moment.  The host keeps a version key,
// but a caller of the reference's era writes the table through `.data` (embeddings.py:36-39 assigns `embeds.data`; torch-1.5
// optimisers do `p.data.add_()`), which moves no version counter.  So validity is decided where the bytes are: one kernel sums a
// 64-bit position-weighted digest of the table (HBM / Infinity-Cache read of the table, nothing else), its last block compares it
// with the digest the pack was made from and writes a `changed` word; the pack kernel that follows on the same stream returns at
// once when the word is 0.  No host synchronisation, graph-capturable: a replayed graph repacks by itself after an optimiser step.
//
// state (caller-owned, SYMPA_DIGEST_STATE_BYTES = 4 096 bytes of device memory, zero-initialised once): u64 stored digest | u64 unused |
// u32 block counter, u32 unused | u32 changed, u32 number of changes seen | up to 508 per-block partial sums.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sympa_hip {

constexpr int DIGEST_BLOCK = 1024;
constexpr int DIGEST_MAX_BLOCKS = 508;     // partial sums behind the four state words
constexpr int DIGEST_STATE_BYTES = 32 + 8 * DIGEST_MAX_BLOCKS;     // = SYMPA_DIGEST_STATE_BYTES (4 096)
constexpr int DIGEST_GUARD_WORD = 6;       // index of the `changed` word in the state seen as u32[8]


// enqueues the digest of `bytes` bytes (a multiple of 8, 16-byte aligned) at `data`; afterwards (stream order)
// ((const uint32_t*)state)[DIGEST_GUARD_WORD] is 1 when the bytes differ from those of the previous call (or force), else 0
int launch_table_digest(const void* data, int64_t bytes, void* state, int force, hipStream_t s);

}  // namespace sympa_hip
