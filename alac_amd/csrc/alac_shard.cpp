// alac_shard.cpp — the rank arithmetic of a sharded encode, host only: which units a rank takes and
// where its shard lands in the re-assembled stream.  Packets are byte aligned (codec/ALACEncoder.cu:1039 in the
// reference), so re-assembly is pure byte placement at the prefix sums computed here; the bytes themselves move with
// RCCL (alac_amd/reassemble.py: grouped send/receive straight to these offsets).
#include "alac_hip.h"

static const int32_t kParamError = -50;  // kALAC_ParamError

extern "C" {

int32_t alac_hip_shard_range(uint64_t num_units, uint32_t world, uint32_t rank, uint64_t *first, uint64_t *count)
{
    if (world == 0 || rank >= world || !first || !count) return kParamError;
    // contiguous ranges, the remainder spread over the first ranks: rank r gets [r S / G, (r + 1) S / G) rounded so that
    // the ranges tile [0, S) exactly
    const uint64_t base = num_units / world, extra = num_units % world;
    *first = (uint64_t)rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
    return 0;
}

int32_t alac_hip_shard_offsets(const uint64_t *shard_bytes, uint32_t world, uint64_t *offsets)
{
    if (world == 0 || !shard_bytes || !offsets) return kParamError;
    uint64_t at = 0;
    for (uint32_t r = 0; r < world; r++) {
        offsets[r] = at;
        if (shard_bytes[r] > UINT64_MAX - at) return kParamError;
        at += shard_bytes[r];
    }
    offsets[world] = at;
    return 0;
}

}  // extern "C"
