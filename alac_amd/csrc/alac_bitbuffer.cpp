// alac_bitbuffer.cpp — the BitBuffer routines of the reference's public surface (codec/ALACBitUtilities.h:85-97,
// behaviour of codec/ALACBitUtilities.c:32-260), exported from libalac_hip.so with the same prototypes.  Pure host
// pointer work on the caller's packet buffer: this is the cursor ALACDecoder::Decode receives and the writer third-party
// code uses for headers; the bulk bit packing of the encode path happens on the device (k_pack).
//
// Written around one primitive, the 24-bit window at the cursor: the three bytes at cur, shifted left by bitIndex and
// cut to 24 bits.  Like the reference the readers look at cur[1] / cur[2] whatever numBits is, so a buffer needs two
// bytes of slack behind the last bit read.
#include "alac/ALACBitUtilities.h"

namespace {

inline uint32_t window24(const BitBuffer *b)
{
    const uint32_t w = ((uint32_t)b->cur[0] << 16) | ((uint32_t)b->cur[1] << 8) | (uint32_t)b->cur[2];
    return (w << b->bitIndex) & 0x00ffffffu;
}

inline void step(BitBuffer *b, uint32_t nbits)
{
    const uint32_t pos = b->bitIndex + nbits;
    b->cur += pos >> 3;
    b->bitIndex = pos & 7u;
}

inline uint8_t *origin(const BitBuffer *b) { return b->end - b->byteSize; }

}  // namespace

extern "C" {

void BitBufferInit(BitBuffer *bits, uint8_t *buffer, uint32_t byteSize)
{
    bits->cur = buffer;
    bits->end = buffer + byteSize;
    bits->bitIndex = 0;
    bits->byteSize = byteSize;
}

// <= 16 bits (what fits the 24-bit window at any bitIndex)
uint32_t BitBufferRead(BitBuffer *bits, uint8_t numBits)
{
    const uint32_t v = numBits ? window24(bits) >> (24u - numBits) : 0u;
    step(bits, numBits);
    return v;
}

// <= 8 bits, from a 16-bit window
uint8_t BitBufferReadSmall(BitBuffer *bits, uint8_t numBits)
{
    const uint16_t w = (uint16_t)((((uint32_t)bits->cur[0] << 8) | bits->cur[1]) << bits->bitIndex);
    const uint8_t v = numBits ? (uint8_t)(w >> (16u - numBits)) : (uint8_t)0;
    step(bits, numBits);
    return v;
}

uint8_t BitBufferReadOne(BitBuffer *bits)
{
    const uint8_t v = (uint8_t)((bits->cur[0] >> (7u - bits->bitIndex)) & 1u);
    step(bits, 1);
    return v;
}

uint32_t BitBufferPeek(BitBuffer *bits, uint8_t numBits) { return numBits ? window24(bits) >> (24u - numBits) : 0u; }

uint32_t BitBufferPeekOne(BitBuffer *bits) { return (uint32_t)((bits->cur[0] >> (7u - bits->bitIndex)) & 1u); }

// BER: 7 payload bits per byte, most significant group first, bit 7 = "more follows"
uint32_t BitBufferUnpackBERSize(BitBuffer *bits)
{
    uint32_t size = 0;
    uint8_t byte;
    do {
        byte = BitBufferReadSmall(bits, 8);
        size = (size << 7) | (byte & 0x7fu);
    } while (byte & 0x80u);
    return size;
}

uint32_t BitBufferGetPosition(BitBuffer *bits) { return (uint32_t)(bits->cur - origin(bits)) * 8u + bits->bitIndex; }

void BitBufferByteAlign(BitBuffer *bits, int32_t addZeros)
{
    if (bits->bitIndex == 0) return;
    const uint32_t pad = 8u - bits->bitIndex;
    if (addZeros)
        BitBufferWrite(bits, 0, pad);
    else
        step(bits, pad);
}

void BitBufferAdvance(BitBuffer *bits, uint32_t numBits) { step(bits, numBits); }

// back by numBits; a rewind past the buffer's first byte ends at its bit 0
void BitBufferRewind(BitBuffer *bits, uint32_t numBits)
{
    uint8_t *first = origin(bits);
    const int64_t pos = (int64_t)(bits->cur - first) * 8 + bits->bitIndex - (int64_t)numBits;
    const uint64_t np = pos > 0 ? (uint64_t)pos : 0u;
    bits->cur = first + (np >> 3);
    bits->bitIndex = (uint32_t)(np & 7u);
}

// MSB-first write of the low numBits of value; bits of the touched bytes outside the field are kept
void BitBufferWrite(BitBuffer *bits, uint32_t value, uint32_t numBits)
{
    if (!bits || numBits == 0) return;
    uint8_t *q = bits->cur;
    uint32_t bi = bits->bitIndex, left = numBits;
    while (left) {
        const uint32_t room = 8u - bi;
        const uint32_t take = left < room ? left : room;
        const uint32_t field = (1u << take) - 1u;
        const uint32_t sh = room - take;
        const uint32_t piece = (value >> (left - take)) & field;
        *q = (uint8_t)((*q & ~(field << sh)) | (piece << sh));
        left -= take;
        bi += take;
        q += bi >> 3;
        bi &= 7u;
    }
    bits->cur = q;
    bits->bitIndex = bi;
}

void BitBufferReset(BitBuffer *bits)
{
    bits->cur = origin(bits);
    bits->bitIndex = 0;
}

}  // extern "C"
