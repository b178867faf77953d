// alac_stage_compat.cpp — host-callable stage functions with the reference's own prototypes
// (codec/dplib.h:49-55, codec/aglib.h:70-74) for third-party code that links them; the arithmetic runs on the
// GPU through the batched stage-level entry points of alac_hip.h as a one-row batch.  Meant for drop-in linking
// and spot checks, not for throughput: every call is a host -> device -> host round trip.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "alac/aglib.h"
#include "alac/dplib.h"
#include "alac_hip.h"

namespace {

alac_hip_ctx *compat_ctx()
{
    static alac_hip_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        if (alac_hip_create(&ctx, 0, nullptr) != ALAC_HIP_noErr) ctx = nullptr;
    });
    return ctx;
}

struct Dev {
    void *p = nullptr;
    explicit Dev(size_t n) { if (hipMalloc(&p, n ? n : 4) != hipSuccess) p = nullptr; }
    ~Dev() { if (p) (void)hipFree(p); }
};

// one row through the batched predictor entry points
void run_pc(bool decode, int32_t *src, int32_t *dst, int32_t num, int16_t *coefs, int32_t numactive, uint32_t chanbits,
            uint32_t denshift)
{
    alac_hip_ctx *ctx = compat_ctx();
    if (!ctx || num <= 0) return;
    const int32_t na = numactive == 31 ? 1 : numactive;
    const uint32_t stride = (uint32_t)((num > na + 1 ? num : na + 1) + 8);
    Dev dIn(stride * 4), dOut(stride * 4), dCo(32 * 2);
    if (!dIn.p || !dOut.p || !dCo.p) return;
    int16_t co[32] = {0};
    const int ncopy = numactive > 0 && numactive <= 32 ? numactive : 0;
    memcpy(co, coefs, ncopy * 2);
    (void)hipMemset(dIn.p, 0, stride * 4);
    (void)hipMemcpy(dIn.p, src, (size_t)(stride - 8) * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dCo.p, co, sizeof(co), hipMemcpyHostToDevice);
    int32_t rc = decode ? alac_hip_unpc_block(ctx, (const int32_t *)dIn.p, (int32_t *)dOut.p, 1, stride, num,
                                              (int16_t *)dCo.p, numactive, chanbits, denshift)
                        : alac_hip_pc_block(ctx, (const int32_t *)dIn.p, (int32_t *)dOut.p, 1, stride, num, (int16_t *)dCo.p,
                                            numactive, chanbits, denshift);
    if (rc != ALAC_HIP_noErr) return;
    alac_hip_synchronize(ctx);
    (void)hipMemcpy(dst, dOut.p, (size_t)num * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(co, dCo.p, sizeof(co), hipMemcpyDeviceToHost);
    memcpy(coefs, co, ncopy * 2);
}

}  // namespace

extern "C" {

// codec/dp_enc.c:49-67
void init_coefs(int16_t *coefs, uint32_t denshift, int32_t numPairs)
{
    const int32_t den = 1 << denshift;
    coefs[0] = (int16_t)((AINIT * den) >> 4);
    coefs[1] = (int16_t)((BINIT * den) >> 4);
    coefs[2] = (int16_t)((CINIT * den) >> 4);
    for (int32_t k = 3; k < numPairs; k++) coefs[k] = 0;
}

void copy_coefs(int16_t *srcCoefs, int16_t *dstCoefs, int32_t numPairs)
{
    for (int32_t k = 0; k < numPairs; k++) dstCoefs[k] = srcCoefs[k];
}

void pc_block(int32_t *in, int32_t *pc, int32_t num, int16_t *coefs, int32_t numactive, uint32_t chanbits, uint32_t denshift)
{
    run_pc(false, in, pc, num, coefs, numactive, chanbits, denshift);
}

void unpc_block(int32_t *pc, int32_t *out, int32_t num, int16_t *coefs, int32_t numactive, uint32_t chanbits,
                uint32_t denshift)
{
    run_pc(true, pc, out, num, coefs, numactive, chanbits, denshift);
}

// codec/ag_dec.c:62-83
void set_ag_params(AGParamRecPtr params, uint32_t m, uint32_t p, uint32_t k, uint32_t f, uint32_t s, uint32_t maxrun)
{
    params->mb = params->mb0 = m;
    params->pb = p;
    params->kb = k;
    params->wb = (1u << params->kb) - 1;
    params->qb = QB - params->pb;
    params->fw = f;
    params->sw = s;
    params->maxrun = maxrun;
}

void set_standard_ag_params(AGParamRecPtr params, uint32_t fullwidth, uint32_t sectorwidth)
{
    set_ag_params(params, MB0, PB0, KB0, fullwidth, sectorwidth, MAX_RUN_DEFAULT);
}

// codec/ag_enc.c:249-367: codes numSamples residuals at the BitBuffer's position and advances it
int32_t dyn_comp(AGParamRecPtr params, int32_t *pc, BitBuffer *bitstream, int32_t numSamples, int32_t bitSize,
                 uint32_t *outNumBits)
{
    if (outNumBits) *outNumBits = 0;
    if (!params || !pc || !bitstream || !outNumBits || bitSize < 1 || bitSize > 32) return kALAC_ParamError;
    alac_hip_ctx *ctx = compat_ctx();
    if (!ctx) return kALAC_ParamError;
    if (numSamples <= 0) return ALAC_noErr;
    const uint32_t cap = ((uint32_t)numSamples * (9u + (uint32_t)bitSize + 25u) + 7) / 8 + 16;
    Dev dPc((size_t)numSamples * 4), dBits(cap), dNum(4);
    if (!dPc.p || !dBits.p || !dNum.p) return kALAC_MemFullError;
    (void)hipMemcpy(dPc.p, pc, (size_t)numSamples * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dBits.p, 0, cap);
    int32_t rc = alac_hip_dyn_comp(ctx, params->mb0, params->pb, params->kb, (const int32_t *)dPc.p, 1, (uint32_t)numSamples,
                                   numSamples, bitSize, (uint8_t *)dBits.p, cap, (uint32_t *)dNum.p);
    if (rc != ALAC_HIP_noErr) return rc;
    alac_hip_synchronize(ctx);
    uint32_t nbits = 0;
    (void)hipMemcpy(&nbits, dNum.p, 4, hipMemcpyDeviceToHost);
    std::vector<uint8_t> bits((nbits + 7) / 8 + 1, 0);
    (void)hipMemcpy(bits.data(), dBits.p, (nbits + 7) / 8, hipMemcpyDeviceToHost);
    // splice at (cur, bitIndex), MSB first; the reference ORs into the buffer the same way (dyn_jam_noDeref)
    uint8_t *out = bitstream->cur;
    uint32_t bi = bitstream->bitIndex;
    for (uint32_t b = 0; b < nbits; b++) {
        const uint32_t v = (bits[b >> 3] >> (7 - (b & 7))) & 1u;
        const uint32_t at = bi + b;
        uint8_t &dst = out[at >> 3];
        const uint8_t mask = (uint8_t)(0x80u >> (at & 7));
        dst = v ? (uint8_t)(dst | mask) : (uint8_t)(dst & ~mask);
    }
    const uint32_t end = bi + nbits;
    bitstream->cur += end >> 3;
    bitstream->bitIndex = end & 7;
    *outNumBits = nbits;
    params->mb = params->mb0;
    return ALAC_noErr;
}

// codec/ag_dec.c:272-362
int32_t dyn_decomp(AGParamRecPtr params, BitBuffer *bitstream, int32_t *pc, int32_t numSamples, int32_t maxSize,
                   uint32_t *outNumBits)
{
    if (outNumBits) *outNumBits = 0;
    if (!params || !bitstream || !pc || !outNumBits) return kALAC_ParamError;
    alac_hip_ctx *ctx = compat_ctx();
    if (!ctx) return kALAC_ParamError;
    if (numSamples <= 0) return ALAC_noErr;
    // bit-align the input to the BitBuffer's position (host splice), hand over what is left of the buffer
    const uint8_t *in = bitstream->cur;
    const uint32_t bi = bitstream->bitIndex;
    const size_t avail = bitstream->end > bitstream->cur ? (size_t)(bitstream->end - bitstream->cur) : 0;
    std::vector<uint8_t> shifted(avail + 8, 0);
    for (size_t i = 0; i < avail; i++) {
        const uint32_t two = ((uint32_t)in[i] << 8) | (i + 1 < avail ? in[i + 1] : 0u);
        shifted[i] = (uint8_t)(two >> (8 - bi));
    }
    const uint32_t nbytes = (uint32_t)avail;
    Dev dBits(nbytes + 16), dPc((size_t)numSamples * 4), dNum(4), dSt(4);
    if (!dBits.p || !dPc.p || !dNum.p || !dSt.p) return kALAC_MemFullError;
    (void)hipMemset(dBits.p, 0, nbytes + 16);
    (void)hipMemcpy(dBits.p, shifted.data(), nbytes, hipMemcpyHostToDevice);
    int32_t rc = alac_hip_dyn_decomp(ctx, params->mb0, params->pb, params->kb, (const uint8_t *)dBits.p, nbytes + 16, 1,
                                     (int32_t *)dPc.p, (uint32_t)numSamples, numSamples, maxSize, (uint32_t *)dNum.p,
                                     (int32_t *)dSt.p);
    if (rc != ALAC_HIP_noErr) return rc;
    alac_hip_synchronize(ctx);
    uint32_t nbits = 0;
    int32_t status = 0;
    (void)hipMemcpy(&nbits, dNum.p, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&status, dSt.p, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(pc, dPc.p, (size_t)numSamples * 4, hipMemcpyDeviceToHost);
    const uint32_t end = bi + nbits;
    bitstream->cur += end >> 3;
    bitstream->bitIndex = end & 7;
    *outNumBits = nbits;
    params->mb = params->mb0;
    return status;
}

}  // extern "C"
