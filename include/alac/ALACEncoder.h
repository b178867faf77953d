/*
 * ALACEncoder.h — drop-in for the reference's class ALACEncoder (codec/ALACEncoder.h:34-102): same method
 * names, argument meaning and int32 status codes, backed by the HIP path (libalac_hip.so).  The fork's
 * extra `index` / `X` arguments are accepted; the upstream (Apple) signatures are the same calls with the
 * defaults.  No device pointers are exposed (the fork's six public d_/dev_ members are gone).
 */
#ifndef ALAC_AMD_ENCODER_H
#define ALAC_AMD_ENCODER_H

#include <stdint.h>
#include <vector>

#include "ALACAudioTypes.h"

struct alac_hip_ctx;

class ALACEncoder {
public:
    ALACEncoder();
    virtual ~ALACEncoder();

    /* codec/ALACEncoder.h:40-41.  *ioNumBytes in = PCM bytes of this packet, out = packet bytes.
     * After InitializeSampling(..., X, ...) packet `index` was already encoded in the batch and is just
     * copied out; otherwise the packet is encoded now (chained to the previous call's state). */
    virtual int32_t Encode(AudioFormatDescription theInputFormat, AudioFormatDescription theOutputFormat,
                           unsigned char *theReadBuffer, unsigned char *theWriteBuffer, int32_t *ioNumBytes,
                           int index = -1);
    virtual int32_t Finish();

    void SetFastMode(bool fast) { mFastMode = fast; }
    void SetFrameSize(uint32_t frameSize) { mFrameSize = frameSize; } /* before InitializeEncoder */
    /* Extension: the HIP device this object's context is created on (before InitializeEncoder; default: environment
     * ALAC_HIP_DEVICE, else 0).  One object = one device = one stream; objects on different devices may be driven from
     * different threads (alacconvert --batch --devices N). */
    void SetDevice(int device) { mDevice = device; }

    void GetConfig(ALACSpecificConfig &config);
    uint32_t GetMagicCookieSize(uint32_t inNumChannels);
    void GetMagicCookie(void *config, uint32_t *ioSize);
    virtual int32_t InitializeEncoder(AudioFormatDescription theOutputFormat, int X = 0);

    /* codec/ALACEncoder.h:53 / ALACEncoder.cu:1385: d_ip = DEVICE buffer holding X packets at a stride of
     * outBytes[0] bytes, outBytes[i] = PCM bytes of packet i.  The whole chained encode of the X packets
     * runs here on the GPU; Encode(index) then returns packet `index`. */
    void InitializeSampling(void *d_ip, AudioFormatDescription theInputFormat, int X, int32_t *outBytes);

    /* Batch extension (host buffers): totalSamples sample-frames of packed LE PCM; segmentPackets = 0
     * chains everything to this object's state, k > 0 restarts from init_coefs every k packets. */
    int32_t EncodeBatch(const void *pcm, uint64_t totalSamples, uint32_t segmentPackets, uint8_t *out,
                        uint64_t outCapacity, uint32_t *packetBytes, uint64_t *outTotalBytes);

    /* Multi-stream form: packets back to back at the full-packet stride (frameSize * bytesPerFrame), packet p
     * holding numSamples[p] frames; segment s = packets [segFirst[s], segFirst[s+1]) is one independent stream
     * (one input file) starting from the initial coefficient state. */
    int32_t EncodeSegments(const void *pcm, const uint32_t *numSamples, uint32_t numPackets, const uint32_t *segFirst,
                           uint32_t numSegments, uint8_t *out, uint64_t outCapacity, uint32_t *packetBytes,
                           uint64_t *outTotalBytes);

    int32_t LastStatus() const { return mLastStatus; }

protected:
    int16_t mBitDepth;
    bool mFastMode;
    uint32_t mTotalBytesGenerated, mAvgBitRate, mMaxFrameBytes;
    uint32_t mFrameSize, mMaxOutputBytes, mNumChannels, mOutputSampleRate;

private:
    alac_hip_ctx *mCtx;
    int mDevice; /* -1: ALAC_HIP_DEVICE or 0 */
    int16_t mState[64 * 8]; /* rows 3 and 7 of mCoefsU/V[first channel of every element] (codec/ALACEncoder.h:89-90) */
    bool mStateValid;
    std::vector<uint8_t> mBatchStream;
    std::vector<uint32_t> mBatchSizes;
    std::vector<uint64_t> mBatchOffsets;
    int32_t mLastStatus;
    void account(uint32_t outputSize);
};

#endif
