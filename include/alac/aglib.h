/*
 * aglib.h — the reference's adaptive-Golomb stage surface (codec/aglib.h:40-74), same prototypes, host
 * pointers.  dyn_comp / dyn_decomp run on the GPU (one-row batch through alac_hip_dyn_comp /
 * alac_hip_dyn_decomp, device 0) and splice the bits into / out of the caller's BitBuffer at its current
 * position, advancing it as the reference does.
 */
#ifndef ALAC_AMD_AGLIB_H
#define ALAC_AMD_AGLIB_H
#include <stdint.h>
#include "ALACAudioTypes.h"
#ifdef __cplusplus
extern "C" {
#endif
#define QBSHIFT 9
#define QB (1 << QBSHIFT)
#define PB0 40
#define MB0 10
#define KB0 14
#define MAX_RUN_DEFAULT 255
typedef struct AGParamRec {
    uint32_t mb, mb0, pb, kb, wb, qb;
    uint32_t fw, sw;
    uint32_t maxrun;
} AGParamRec, *AGParamRecPtr;
void set_standard_ag_params(AGParamRecPtr params, uint32_t fullwidth, uint32_t sectorwidth);
void set_ag_params(AGParamRecPtr params, uint32_t m, uint32_t p, uint32_t k, uint32_t f, uint32_t s, uint32_t maxrun);
int32_t dyn_comp(AGParamRecPtr params, int32_t *pc, struct BitBuffer *bitstream, int32_t numSamples, int32_t bitSize,
                 uint32_t *outNumBits);
int32_t dyn_decomp(AGParamRecPtr params, struct BitBuffer *bitstream, int32_t *pc, int32_t numSamples, int32_t maxSize,
                   uint32_t *outNumBits);
#ifdef __cplusplus
}
#endif
#endif
