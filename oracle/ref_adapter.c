/*
 * ref_adapter.c — TEST INFRASTRUCTURE ONLY.
 *
 * Thin flat-signature shims over the REFERENCE's own compiled stage objects so that the
 * oracle's drivers (oalac_hooks) and the stage-level tests can call them.  Built only where
 * /root/reference exists (this container); includes the reference headers from where they lie,
 * copies nothing.  Output goes to oracle/_ref/ (git-ignored, travels with gpurun).
 */
#include <stdint.h>
#include <string.h>

#include "aglib.h"             /* /root/reference/codec/aglib.h:57-74 */
#include "dplib.h"             /* /root/reference/codec/dplib.h:49-55 */
#include "ALACBitUtilities.h"  /* /root/reference/codec/ALACBitUtilities.h:71-97 */

static void cursor_in(BitBuffer *bb, uint8_t *buf, uint64_t bufbytes, uint64_t bitpos)
{
    bb->cur = buf + (bitpos >> 3);
    bb->end = buf + bufbytes;
    bb->bitIndex = (uint32_t)(bitpos & 7);
    bb->byteSize = (uint32_t)bufbytes;
}

/* ag_enc.c:249 dyn_comp through a bit-position cursor; buf needs >= 8 bytes of slack */
int32_t ref_dyn_comp_flat(uint32_t mb0, uint32_t pb, uint32_t kb, int32_t *pc, uint8_t *buf,
                          uint64_t *bitpos, int32_t numSamples, int32_t bitSize, uint32_t *outNumBits)
{
    AGParamRec p;
    BitBuffer bb;
    set_ag_params(&p, mb0, pb, kb, (uint32_t)numSamples, (uint32_t)numSamples, MAX_RUN_DEFAULT);
    cursor_in(&bb, buf, 0x7fffffffu, *bitpos);
    int32_t st = dyn_comp(&p, pc, &bb, numSamples, bitSize, outNumBits);
    *bitpos = (uint64_t)(bb.cur - buf) * 8 + bb.bitIndex;
    return st;
}

/* ag_dec.c:272 dyn_decomp; buf needs >= 8 readable bytes of slack past bufbytes */
int32_t ref_dyn_decomp_flat(uint32_t mb0, uint32_t pb, uint32_t kb, uint8_t *buf, uint64_t bufbytes,
                            uint64_t *bitpos, int32_t *pc, int32_t numSamples, int32_t maxSize,
                            uint32_t *outNumBits)
{
    AGParamRec p;
    BitBuffer bb;
    set_ag_params(&p, mb0, pb, kb, (uint32_t)numSamples, (uint32_t)numSamples, MAX_RUN_DEFAULT);
    cursor_in(&bb, buf, bufbytes, *bitpos);
    int32_t st = dyn_decomp(&p, &bb, pc, numSamples, maxSize, outNumBits);
    *bitpos = (uint64_t)(bb.cur - buf) * 8 + bb.bitIndex;
    return st;
}

/* ALACBitUtilities.c:212 / :42 through a bit-position cursor */
void ref_put_bits(uint8_t *buf, uint64_t *bitpos, uint32_t value, uint32_t numBits)
{
    BitBuffer bb;
    cursor_in(&bb, buf, 0x7fffffffu, *bitpos);
    BitBufferWrite(&bb, value, numBits);
    *bitpos = (uint64_t)(bb.cur - buf) * 8 + bb.bitIndex;
}

uint32_t ref_get_bits16(uint8_t *buf, uint64_t *bitpos, uint32_t numBits)
{
    BitBuffer bb;
    cursor_in(&bb, buf, 0x7fffffffu, *bitpos);
    uint32_t v = BitBufferRead(&bb, (uint8_t)numBits);
    *bitpos = (uint64_t)(bb.cur - buf) * 8 + bb.bitIndex;
    return v;
}
