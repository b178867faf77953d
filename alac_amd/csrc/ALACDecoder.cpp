// ALACDecoder.cpp — host C++ mirror of the reference's ALACDecoder (codec/ALACDecoder.cu) over the
// alac_hip C-ABI.
#include "alac/ALACDecoder.h"
#include "alac_hip.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

namespace {
inline uint32_t bps_of(uint32_t depth) { return depth == 16 ? 2u : (depth == 32 ? 4u : 3u); }

// sample count of the first audio element of a packet (header peek, codec/ALACDecoder.cu:621-654)
uint32_t peek_num_samples(const uint8_t *p, size_t n, uint32_t frameLength)
{
    if (n < 3) return 0;
    const uint32_t tag = p[0] >> 5;
    if (!(tag == 0 || tag == 1 || tag == 3)) return frameLength;
    // bits: 3 tag, 4 instance, 12 unused, 4 flags -> the partial flag is bit 19
    const uint32_t flags = ((p[2] >> 1) & 0xf);
    if (!(flags & 8)) return frameLength;
    if (n < 7) return 0;
    uint64_t v = 0;
    for (int i = 2; i < 7; i++) v = (v << 8) | p[i];
    return (uint32_t)((v >> 1) & 0xffffffffu);  // 32 bits starting at bit 23
}
}  // namespace

ALACDecoder::ALACDecoder() : mCtx(nullptr), mDevice(-1), mLastStatus(0) { memset(&mConfig, 0, sizeof(mConfig)); }

ALACDecoder::~ALACDecoder()
{
    if (mCtx) alac_hip_destroy(mCtx);
}

int32_t ALACDecoder::Init(void *inMagicCookie, uint32_t inMagicCookieSize, int /*X*/)
{
    alac_hip_format fmt;
    if (!inMagicCookie) return kALAC_ParamError;
    int32_t rc = alac_hip_format_from_cookie((const uint8_t *)inMagicCookie, inMagicCookieSize, &fmt);
    if (rc != ALAC_HIP_noErr) return kALAC_ParamError;
    const uint8_t *ck = (const uint8_t *)inMagicCookie;
    uint32_t size = inMagicCookieSize;
    if (size >= 12 && ck[4] == 'f' && ck[5] == 'r' && ck[6] == 'm' && ck[7] == 'a') { ck += 12; size -= 12; }
    if (size >= 12 && ck[4] == 'a' && ck[5] == 'l' && ck[6] == 'a' && ck[7] == 'c') { ck += 12; size -= 12; }
    mCookie.assign(ck, ck + 24);
    mConfig.frameLength = fmt.frame_size;
    mConfig.compatibleVersion = ck[4];
    mConfig.bitDepth = ck[5];
    mConfig.pb = ck[6];
    mConfig.mb = ck[7];
    mConfig.kb = ck[8];
    mConfig.numChannels = ck[9];
    mConfig.maxRun = (uint16_t)((ck[10] << 8) | ck[11]);
    mConfig.maxFrameBytes = ((uint32_t)ck[12] << 24) | ((uint32_t)ck[13] << 16) | ((uint32_t)ck[14] << 8) | ck[15];
    mConfig.avgBitRate = ((uint32_t)ck[16] << 24) | ((uint32_t)ck[17] << 16) | ((uint32_t)ck[18] << 8) | ck[19];
    mConfig.sampleRate = fmt.sample_rate;
    if (!mCtx) {
        const char *dev = getenv("ALAC_HIP_DEVICE");
        if (alac_hip_create(&mCtx, mDevice >= 0 ? mDevice : (dev ? atoi(dev) : 0), nullptr) != ALAC_HIP_noErr) return kALAC_MemFullError;
    }
    mQueued.clear();
    mQueuedSizes.clear();
    return ALAC_noErr;
}

int32_t ALACDecoder::DecodeBatch(const uint8_t *stream, const uint32_t *packetBytes, uint32_t numPackets,
                                 uint8_t *pcmOut, uint32_t *numSamplesOut, int32_t *statusOut)
{
    if (!mCtx || mCookie.empty()) return kALAC_ParamError;
    mLastStatus = alac_hip_decode_host(mCtx, mCookie.data(), (uint32_t)mCookie.size(), stream, packetBytes, numPackets,
                                       pcmOut, numSamplesOut, statusOut);
    return mLastStatus;
}

int32_t ALACDecoder::Decode(BitBuffer *bits, uint8_t *sampleBuffer, uint32_t /*numSamples*/, uint32_t numChannels,
                            uint32_t *outNumSamples)
{
    if (!bits || !sampleBuffer || !outNumSamples || numChannels == 0) return kALAC_ParamError;
    if (!(bits->cur < bits->end)) return kALAC_ParamError;  // :615
    const uint32_t n = (uint32_t)(bits->end - bits->cur);
    const uint32_t bpf = mConfig.numChannels * bps_of(mConfig.bitDepth);
    std::vector<uint8_t> pcm((size_t)mConfig.frameLength * bpf);
    uint32_t ns = 0;
    int32_t st = 0;
    int32_t rc = DecodeBatch(bits->cur, &n, 1, pcm.data(), &ns, &st);
    if (rc != ALAC_noErr) return rc;
    if (st != 0) return st;
    memcpy(sampleBuffer, pcm.data(), (size_t)ns * bpf);
    *outNumSamples = ns;
    bits->cur = bits->end;  // the whole packet was consumed
    bits->bitIndex = 0;
    return ALAC_noErr;
}

int32_t ALACDecoder::Decode(BitBuffer *bits, uint32_t numSamples, uint32_t numChannels, uint32_t *outNumSamples,
                            uint32_t /*outBytesPerPacket*/, int X)
{
    if (!bits || !outNumSamples || numChannels == 0 || X < 0) return kALAC_ParamError;
    if (!(bits->cur < bits->end)) return kALAC_ParamError;
    if ((size_t)X != mQueuedSizes.size()) return kALAC_ParamError;  // packets are queued in order
    const size_t n = (size_t)(bits->end - bits->cur);
    mQueued.insert(mQueued.end(), bits->cur, bits->end);
    mQueuedSizes.push_back((uint32_t)n);
    uint32_t ns = peek_num_samples(bits->cur, n, mConfig.frameLength);
    *outNumSamples = ns ? ns : numSamples;
    return ALAC_noErr;
}

void ALACDecoder::fillWriteBuffer(void *deviceSampleBuffer, uint32_t /*numChannels*/, int32_t theOutputPacketBytes,
                                  int /*X*/)
{
    mLastStatus = kALAC_ParamError;
    if (!mCtx || mCookie.empty() || !deviceSampleBuffer || mQueuedSizes.empty()) return;
    const uint32_t bpf = mConfig.numChannels * bps_of(mConfig.bitDepth);
    if ((uint32_t)theOutputPacketBytes != mConfig.frameLength * bpf) return;
    const uint32_t np = (uint32_t)mQueuedSizes.size();
    std::vector<uint64_t> offs(np + 1, 0);
    for (uint32_t i = 0; i < np; i++) offs[i + 1] = offs[i] + mQueuedSizes[i];
    alac_hip_format fmt = {mConfig.frameLength, mConfig.bitDepth, mConfig.numChannels, mConfig.sampleRate};
    const uint64_t wsBytes = alac_hip_decode_workspace_bytes_stream(&fmt, np, offs[np]);
    void *dStream = nullptr, *dOffs = nullptr, *dWs = nullptr, *dNs = nullptr, *dSt = nullptr;
    hipStream_t st = (hipStream_t)alac_hip_stream(mCtx);
    bool ok = hipMalloc(&dStream, offs[np] + 16) == hipSuccess && hipMalloc(&dOffs, (np + 1) * 8ull) == hipSuccess &&
              hipMalloc(&dWs, wsBytes) == hipSuccess && hipMalloc(&dNs, np * 4ull) == hipSuccess &&
              hipMalloc(&dSt, np * 4ull) == hipSuccess;
    if (ok)
        ok = hipMemcpyAsync(dStream, mQueued.data(), offs[np], hipMemcpyHostToDevice, st) == hipSuccess &&
             hipMemcpyAsync(dOffs, offs.data(), (np + 1) * 8ull, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
        mLastStatus = alac_hip_decode(mCtx, mCookie.data(), (uint32_t)mCookie.size(), (const uint8_t *)dStream,
                                      (const uint64_t *)dOffs, np, dWs, wsBytes, (uint8_t *)deviceSampleBuffer,
                                      (uint32_t *)dNs, (int32_t *)dSt);
        if (mLastStatus == ALAC_HIP_noErr && alac_hip_synchronize(mCtx) != ALAC_HIP_noErr) mLastStatus = kALAC_ParamError;
        // per-packet results: a packet that failed to decode must not pass for audio — its slot in the caller's buffer is
        // zeroed and the first failure becomes the status of the call (what Decode returned for that packet in the
        // reference's per-packet loop, convert-utility/main.cu:719-724)
        if (mLastStatus == ALAC_HIP_noErr) {
            std::vector<int32_t> stv(np, 0);
            if (hipMemcpy(stv.data(), dSt, np * 4ull, hipMemcpyDeviceToHost) != hipSuccess) {
                mLastStatus = kALAC_ParamError;
            } else {
                for (uint32_t i = 0; i < np; i++) {
                    if (stv[i] == 0) continue;
                    if (mLastStatus == ALAC_HIP_noErr) mLastStatus = stv[i];
                    (void)hipMemset((uint8_t *)deviceSampleBuffer + (size_t)i * theOutputPacketBytes, 0, (size_t)theOutputPacketBytes);
                }
            }
        }
    } else {
        mLastStatus = kALAC_MemFullError;
    }
    (void)hipFree(dStream); (void)hipFree(dOffs); (void)hipFree(dWs); (void)hipFree(dNs); (void)hipFree(dSt);
    mQueued.clear();
    mQueuedSizes.clear();
}
