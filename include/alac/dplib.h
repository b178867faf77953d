/*
 * dplib.h — the reference's predictor stage surface (codec/dplib.h:49-55), same prototypes, host pointers.
 * pc_block / unpc_block run on the GPU (one-row batch through alac_hip_pc_block / alac_hip_unpc_block of
 * libalac_hip.so, device 0); init_coefs / copy_coefs are plain data initialisation.
 */
#ifndef ALAC_AMD_DPLIB_H
#define ALAC_AMD_DPLIB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define DENSHIFT_MAX 15
#define DENSHIFT_DEFAULT 9
#define AINIT 38
#define BINIT (-29)
#define CINIT (-2)
#define NUMCOEPAIRS 16
void init_coefs(int16_t *coefs, uint32_t denshift, int32_t numPairs);
void copy_coefs(int16_t *srcCoefs, int16_t *dstCoefs, int32_t numPairs);
/* NOTE (as in the reference): these read at least "numactive" samples, the i/o buffers must be that big */
void pc_block(int32_t *in, int32_t *pc, int32_t num, int16_t *coefs, int32_t numactive, uint32_t chanbits, uint32_t denshift);
void unpc_block(int32_t *pc, int32_t *out, int32_t num, int16_t *coefs, int32_t numactive, uint32_t chanbits, uint32_t denshift);
#ifdef __cplusplus
}
#endif
#endif
