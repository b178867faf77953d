/*
 * matrixlib.h — the reference's stereo-matrixing surface (codec/matrixlib.h:41-60): same prototypes.
 * u = (mixres*L + (2^mixbits - mixres)*R) >> mixbits, v = L - R (mixres != 0); u = L, v = R (mixres == 0);
 * the 24/32-bit forms first move the low `bytesShifted` bytes of L and R into shiftUV[2j], shiftUV[2j+1].
 *
 * The arithmetic runs on the GPU (alac_amd/csrc/alac_matrix.hip, one thread per sample-frame).  The buffers may
 * be DEVICE pointers — what the fork's kernels-behind-mixNN take (codec/matrix_enc.cu:101-425, callers
 * codec/ALACEncoder.cu:385-415) — or HOST pointers — Apple's upstream signature; host buffers are staged
 * through the device (a round trip per call: for drop-in linking and spot checks, the hot path fuses the mix
 * into the predictor's staging).  All buffers of one call must be of the same kind.  `stride` is the number of
 * interleaved channels of `in` (sample-frame j starts at element j * stride).
 */
#ifndef ALAC_AMD_MATRIXLIB_H
#define ALAC_AMD_MATRIXLIB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void mix16(int16_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres);
void mix20(uint8_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres);
void mix24(uint8_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres,
           uint16_t *shiftUV, int32_t bytesShifted);
void mix32(int32_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres,
           uint16_t *shiftUV, int32_t bytesShifted);
/* 20-bit samples (left-justified in 3 bytes) of one channel -> right-aligned int32 (the mono path, :812-963) */
void copy20ToPredictor(uint8_t *in, uint32_t stride, int32_t *out, int32_t numSamples);
#ifdef __cplusplus
}
#endif
#endif
