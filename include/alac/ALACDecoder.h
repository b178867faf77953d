/*
 * ALACDecoder.h — drop-in for the reference's class ALACDecoder (codec/ALACDecoder.h:38-72), backed by
 * the HIP path.  Both the upstream (Apple) Decode signature (host sampleBuffer) and the fork's
 * (Decode(..., X) + fillWriteBuffer into a device buffer) are provided.
 */
#ifndef ALAC_AMD_DECODER_H
#define ALAC_AMD_DECODER_H

#include <stdint.h>
#include <vector>

#include "ALACAudioTypes.h"

struct alac_hip_ctx;

class ALACDecoder {
public:
    ALACDecoder();
    ~ALACDecoder();

    /* codec/ALACDecoder.cu:109-190 */
    int32_t Init(void *inMagicCookie, uint32_t inMagicCookieSize, int X = 0);
    /* Extension: the HIP device of this object's context (before Init; default: environment ALAC_HIP_DEVICE, else 0) */
    void SetDevice(int device) { mDevice = device; }

    /* upstream form: decode the packet at bits->cur into host sampleBuffer (packed LE interleaved) */
    int32_t Decode(BitBuffer *bits, uint8_t *sampleBuffer, uint32_t numSamples, uint32_t numChannels,
                   uint32_t *outNumSamples);

    /* fork form (codec/ALACDecoder.h:45-46): queue packet X (bytes bits->cur .. bits->end), report its
     * sample count; fillWriteBuffer decodes every queued packet on the GPU straight into the DEVICE
     * buffer, packet X at X * theOutputPacketBytes. */
    int32_t Decode(BitBuffer *bits, uint32_t numSamples, uint32_t numChannels, uint32_t *outNumSamples,
                   uint32_t outBytesPerPacket, int X);
    void fillWriteBuffer(void *deviceSampleBuffer, uint32_t numChannels, int32_t theOutputPacketBytes, int X);

    /* batch extension (host buffers) */
    int32_t DecodeBatch(const uint8_t *stream, const uint32_t *packetBytes, uint32_t numPackets, uint8_t *pcmOut,
                        uint32_t *numSamplesOut, int32_t *statusOut);

    int32_t LastStatus() const { return mLastStatus; }

public:
    ALACSpecificConfig mConfig;  /* host-endian copy, as in the reference (codec/ALACDecoder.cu:139-151) */

private:
    alac_hip_ctx *mCtx;
    int mDevice; /* -1: ALAC_HIP_DEVICE or 0 */
    std::vector<uint8_t> mCookie;
    std::vector<uint8_t> mQueued;         /* fork form: queued packet bytes */
    std::vector<uint32_t> mQueuedSizes;
    int32_t mLastStatus;
};

#endif
