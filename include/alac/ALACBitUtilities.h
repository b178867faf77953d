/*
 * ALACBitUtilities.h — the reference's bit-buffer surface (codec/ALACBitUtilities.h:51-97): same struct,
 * same prototypes, host pointers.  Implemented in libalac_hip.so (alac_amd/csrc/alac_bitbuffer.cpp); this is
 * the packet cursor ALACDecoder::Decode takes and what third-party code links through alac.pc.
 */
#ifndef ALAC_AMD_BITUTILITIES_H
#define ALAC_AMD_BITUTILITIES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ALAC_AMD_NOERR_DEFINED
#define ALAC_AMD_NOERR_DEFINED
enum { ALAC_noErr = 0 };
#endif

/* element tags of a packet (codec/ALACBitUtilities.h:57-68) */
typedef enum {
    ID_SCE = 0, /* single channel element */
    ID_CPE = 1, /* channel pair element */
    ID_CCE = 2, /* coupling channel element */
    ID_LFE = 3, /* LFE channel element */
    ID_DSE = 4, /* data stream element */
    ID_PCE = 5,
    ID_FIL = 6, /* fill element */
    ID_END = 7
} ELEMENT_TYPE;

typedef struct BitBuffer {
    uint8_t *cur;
    uint8_t *end;
    uint32_t bitIndex;
    uint32_t byteSize;
} BitBuffer;

/* fixed-size buffer cursor; bounds checking is the client's business, as in the reference */
void BitBufferInit(BitBuffer *bits, uint8_t *buffer, uint32_t byteSize);
uint32_t BitBufferRead(BitBuffer *bits, uint8_t numBits); /* <= 16 bits at a time */
uint8_t BitBufferReadSmall(BitBuffer *bits, uint8_t numBits); /* <= 8 bits */
uint8_t BitBufferReadOne(BitBuffer *bits);
uint32_t BitBufferPeek(BitBuffer *bits, uint8_t numBits); /* <= 16 bits */
uint32_t BitBufferPeekOne(BitBuffer *bits);
uint32_t BitBufferUnpackBERSize(BitBuffer *bits);
uint32_t BitBufferGetPosition(BitBuffer *bits);
void BitBufferByteAlign(BitBuffer *bits, int32_t addZeros);
void BitBufferAdvance(BitBuffer *bits, uint32_t numBits);
void BitBufferRewind(BitBuffer *bits, uint32_t numBits);
void BitBufferWrite(BitBuffer *bits, uint32_t value, uint32_t numBits);
void BitBufferReset(BitBuffer *bits);

#ifdef __cplusplus
}
#endif
#endif
