// ALACEncoder.cpp — host C++ mirror of the reference's ALACEncoder (codec/ALACEncoder.cu) over the
// alac_hip C-ABI.  Control flow that lived in EncodeStereo/EncodeMono now lives in the HIP kernels;
// this class only keeps the reference's object model (stateful encoder, cookie, statistics).
#include "alac/ALACEncoder.h"
#include "alac_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
inline uint32_t bps_of(int depth) { return depth == 16 ? 2u : (depth == 32 ? 4u : 3u); }
inline uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }
inline uint16_t be16(uint16_t v) { return (uint16_t)((v << 8) | (v >> 8)); }
}  // namespace

ALACEncoder::ALACEncoder()
    : mBitDepth(0), mFastMode(false), mTotalBytesGenerated(0), mAvgBitRate(0), mMaxFrameBytes(0),
      mFrameSize(kALACDefaultFrameSize), mMaxOutputBytes(0), mNumChannels(0), mOutputSampleRate(0), mCtx(nullptr),
      mDevice(-1), mStateValid(false), mLastStatus(0)
{
    memset(mState, 0, sizeof(mState));
}

ALACEncoder::~ALACEncoder()
{
    if (mCtx) alac_hip_destroy(mCtx);
}

// codec/ALACEncoder.cu:1457-1535
int32_t ALACEncoder::InitializeEncoder(AudioFormatDescription theOutputFormat, int /*X*/)
{
    mOutputSampleRate = (uint32_t)theOutputFormat.mSampleRate;
    mNumChannels = theOutputFormat.mChannelsPerFrame;
    switch (theOutputFormat.mFormatFlags) {
    case 1: mBitDepth = 16; break;
    case 2: mBitDepth = 20; break;
    case 3: mBitDepth = 24; break;
    case 4: mBitDepth = 32; break;
    default: break;
    }
    if (!(mBitDepth == 16 || mBitDepth == 20 || mBitDepth == 24 || mBitDepth == 32)) return kALAC_ParamError;
    if (mNumChannels < 1 || mNumChannels > kALACMaxChannels) return kALAC_ParamError;
    mMaxOutputBytes = mFrameSize * mNumChannels * ((10 + 32) / 8) + 1;        // :1489
    if (!mCtx) {
        const char *dev = getenv("ALAC_HIP_DEVICE");
        int32_t rc = alac_hip_create(&mCtx, mDevice >= 0 ? mDevice : (dev ? atoi(dev) : 0), nullptr);
        if (rc != ALAC_HIP_noErr) return kALAC_MemFullError;
    }
    mStateValid = false;  // every row = init_coefs (:1524-1531)
    mBatchStream.clear();
    mBatchSizes.clear();
    mBatchOffsets.clear();
    return ALAC_noErr;
}

// codec/ALACEncoder.cu:1082-1095 (fields big-endian in the struct)
void ALACEncoder::GetConfig(ALACSpecificConfig &config)
{
    config.frameLength = be32(mFrameSize);
    config.compatibleVersion = (uint8_t)kALACCompatibleVersion;
    config.bitDepth = (uint8_t)mBitDepth;
    config.pb = 40;
    config.kb = 14;
    config.mb = 10;
    config.numChannels = (uint8_t)mNumChannels;
    config.maxRun = be16(255);
    config.maxFrameBytes = be32(mMaxFrameBytes);
    config.avgBitRate = be32(mAvgBitRate);
    config.sampleRate = be32(mOutputSampleRate);
}

// :1097-1107
uint32_t ALACEncoder::GetMagicCookieSize(uint32_t inNumChannels)
{
    return inNumChannels > 2 ? (uint32_t)sizeof(ALACSpecificConfig) + 24u : (uint32_t)sizeof(ALACSpecificConfig);
}

// :1109-1140
void ALACEncoder::GetMagicCookie(void *outCookie, uint32_t *ioSize)
{
    ALACSpecificConfig cfg;
    GetConfig(cfg);
    const uint32_t need = GetMagicCookieSize(mNumChannels);
    if (*ioSize >= need) {
        memcpy(outCookie, &cfg, sizeof(cfg));
        if (need > sizeof(cfg)) {  // 'chan' atom + ALACAudioChannelLayout (:1118-1133)
            uint8_t full[48];
            alac_hip_format fmt = {mFrameSize, (uint32_t)mBitDepth, mNumChannels, mOutputSampleRate};
            alac_hip_magic_cookie_full(&fmt, mMaxFrameBytes, mAvgBitRate, full, sizeof(full));
            memcpy((uint8_t *)outCookie + sizeof(cfg), full + 24, 24);
        }
        *ioSize = need;
    } else {
        *ioSize = 0;  // no incomplete cookies
    }
}

void ALACEncoder::account(uint32_t outputSize)
{
    mTotalBytesGenerated += outputSize;  // :1050-1051
    if (outputSize > mMaxFrameBytes) mMaxFrameBytes = outputSize;
}

int32_t ALACEncoder::EncodeBatch(const void *pcm, uint64_t totalSamples, uint32_t segmentPackets, uint8_t *out,
                                 uint64_t outCapacity, uint32_t *packetBytes, uint64_t *outTotalBytes)
{
    if (!mCtx) return kALAC_ParamError;
    (void)alac_hip_set_option(mCtx, "fast_mode", mFastMode ? 1 : 0);  // SetFastMode: EncodeStereoFast, codec/ALACEncoder.cu:998-1001
    alac_hip_format fmt = {mFrameSize, (uint32_t)mBitDepth, mNumChannels, mOutputSampleRate};
    const uint64_t np = (totalSamples + mFrameSize - 1) / mFrameSize;
    const uint64_t nseg = segmentPackets ? (np + segmentPackets - 1) / segmentPackets : 1;
    const uint32_t stateInt16 = alac_hip_state_int16(&fmt);  // 64 per element of a packet
    std::vector<int16_t> state(nseg * stateInt16, 0);
    const bool chain = (segmentPackets == 0) && mStateValid;
    if (chain) memcpy(state.data(), mState, stateInt16 * 2u);
    uint64_t total = 0;
    mLastStatus = alac_hip_encode_host(mCtx, &fmt, pcm, totalSamples, segmentPackets, state.data(), chain ? 1 : 0, out,
                                       outCapacity, packetBytes, &total);
    if (mLastStatus != ALAC_HIP_noErr) return mLastStatus;
    if (segmentPackets == 0 && np) {
        memcpy(mState, state.data(), stateInt16 * 2u);
        mStateValid = true;
    }
    for (uint64_t p = 0; p < np; p++) account(packetBytes[p]);
    if (outTotalBytes) *outTotalBytes = total;
    return ALAC_noErr;
}

int32_t ALACEncoder::EncodeSegments(const void *pcm, const uint32_t *numSamples, uint32_t numPackets,
                                    const uint32_t *segFirst, uint32_t numSegments, uint8_t *out, uint64_t outCapacity,
                                    uint32_t *packetBytes, uint64_t *outTotalBytes)
{
    if (!mCtx) return kALAC_ParamError;
    (void)alac_hip_set_option(mCtx, "fast_mode", mFastMode ? 1 : 0);
    alac_hip_format fmt = {mFrameSize, (uint32_t)mBitDepth, mNumChannels, mOutputSampleRate};
    uint64_t total = 0;
    mLastStatus = alac_hip_encode_host_segments(mCtx, &fmt, pcm, numSamples, numPackets, segFirst, numSegments, nullptr, 0,
                                                out, outCapacity, packetBytes, &total);
    if (mLastStatus != ALAC_HIP_noErr) return mLastStatus;
    for (uint32_t p = 0; p < numPackets; p++) account(packetBytes[p]);
    if (outTotalBytes) *outTotalBytes = total;
    return ALAC_noErr;
}

// codec/ALACEncoder.cu:1385-1451
void ALACEncoder::InitializeSampling(void *d_ip, AudioFormatDescription theInputFormat, int X, int32_t *outBytes)
{
    mLastStatus = kALAC_ParamError;
    if (!mCtx || X <= 0 || !d_ip || !outBytes) return;
    const uint32_t bpf = theInputFormat.mChannelsPerFrame * bps_of(mBitDepth);
    const int64_t stride = outBytes[0];
    if (stride != (int64_t)mFrameSize * bpf) return;  // first packet must be full, as the fork assumes (:1153)
    // packets the caller really filled: main.cu leaves the tail entries of outBytes[] unset when the file
    // is an exact multiple (X = bytes/packet + 1, convert-utility/main.cu:409) — drop anything implausible
    std::vector<uint32_t> ns;
    for (int i = 0; i < X; i++) {
        const int64_t b = outBytes[i];
        if (b <= 0 || b > stride || (b % bpf) != 0) break;
        ns.push_back((uint32_t)(b / bpf));
        if (b != stride) break;  // a partial packet ends the stream
    }
    const uint32_t np = (uint32_t)ns.size();
    if (np == 0) return;
    (void)alac_hip_set_option(mCtx, "fast_mode", mFastMode ? 1 : 0);
    alac_hip_format fmt = {mFrameSize, (uint32_t)mBitDepth, mNumChannels, mOutputSampleRate};
    const uint32_t segFirst[2] = {0, np};
    const uint64_t stateBytes = alac_hip_state_int16(&fmt) * 2ull;
    const uint64_t wsBytes = alac_hip_encode_workspace_bytes(&fmt, np, 1);
    const uint64_t outMax = alac_hip_encode_max_output_bytes(&fmt, np);
    void *dNs = nullptr, *dSeg = nullptr, *dState = nullptr, *dWs = nullptr, *dOut = nullptr, *dSizes = nullptr,
         *dOffs = nullptr;
    hipStream_t st = (hipStream_t)alac_hip_stream(mCtx);
    bool ok = hipMalloc(&dNs, np * 4ull) == hipSuccess && hipMalloc(&dSeg, 8) == hipSuccess &&
              hipMalloc(&dState, stateBytes) == hipSuccess && hipMalloc(&dWs, wsBytes) == hipSuccess &&
              hipMalloc(&dOut, outMax) == hipSuccess && hipMalloc(&dSizes, np * 4ull) == hipSuccess &&
              hipMalloc(&dOffs, (np + 1) * 8ull) == hipSuccess;
    if (ok) {
        ok = hipMemcpyAsync(dNs, ns.data(), np * 4ull, hipMemcpyHostToDevice, st) == hipSuccess &&
             hipMemcpyAsync(dSeg, segFirst, 8, hipMemcpyHostToDevice, st) == hipSuccess;
        if (ok && mStateValid) ok = hipMemcpyAsync(dState, mState, stateBytes, hipMemcpyHostToDevice, st) == hipSuccess;
    }
    if (ok) {
        // one chained segment of np packets: the bound is known here, so the library does not read the table back
        mLastStatus = alac_hip_encode_segmented(mCtx, &fmt, d_ip, (const uint32_t *)dNs, np, (const uint32_t *)dSeg, 1, np,
                                                (int16_t *)dState, mStateValid ? 1 : 0, dWs, wsBytes, (uint8_t *)dOut, outMax,
                                      (uint32_t *)dSizes, (uint64_t *)dOffs);
        ok = (mLastStatus == ALAC_HIP_noErr);
    } else {
        mLastStatus = kALAC_MemFullError;
    }
    if (ok) {
        mBatchSizes.resize(np);
        mBatchOffsets.resize(np + 1);
        ok = hipMemcpyAsync(mBatchSizes.data(), dSizes, np * 4ull, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipMemcpyAsync(mBatchOffsets.data(), dOffs, (np + 1) * 8ull, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipMemcpyAsync(mState, dState, stateBytes, hipMemcpyDeviceToHost, st) == hipSuccess;
        // alac_hip_synchronize, not a bare stream sync: it also reads the context's hand-off error word — a consumer wave
        // that gave up waiting for its producer has coded garbage, and the batch must fail instead of being handed out
        int32_t syncRc = ALAC_HIP_noErr;
        if (ok) {
            syncRc = alac_hip_synchronize(mCtx);
            ok = syncRc == ALAC_HIP_noErr;
        }
        if (ok) {
            mBatchStream.resize(mBatchOffsets[np]);
            ok = hipMemcpy(mBatchStream.data(), dOut, mBatchOffsets[np], hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (ok) mStateValid = true;
        mLastStatus = ok ? ALAC_noErr : (syncRc != ALAC_HIP_noErr ? syncRc : kALAC_ParamError);
    }
    if (!ok) {
        mBatchStream.clear();
        mBatchSizes.clear();
        mBatchOffsets.clear();
    }
    (void)hipFree(dNs); (void)hipFree(dSeg); (void)hipFree(dState); (void)hipFree(dWs);
    (void)hipFree(dOut); (void)hipFree(dSizes); (void)hipFree(dOffs);
}

// codec/ALACEncoder.cu:973-1057
int32_t ALACEncoder::Encode(AudioFormatDescription theInputFormat, AudioFormatDescription /*theOutputFormat*/,
                            unsigned char *theReadBuffer, unsigned char *theWriteBuffer, int32_t *ioNumBytes,
                            int index)
{
    if (!mCtx || !ioNumBytes || !theWriteBuffer) return kALAC_ParamError;
    if (theInputFormat.mChannelsPerFrame != mNumChannels) return kALAC_ParamError;
    // batch already encoded by InitializeSampling: hand out packet `index`
    if (index >= 0 && (size_t)index < mBatchSizes.size()) {
        const uint32_t n = mBatchSizes[index];
        memcpy(theWriteBuffer, mBatchStream.data() + mBatchOffsets[index], n);
        *ioNumBytes = (int32_t)n;
        account(n);
        return ALAC_noErr;
    }
    if (!theReadBuffer) return kALAC_ParamError;
    const uint32_t bpf = mNumChannels * bps_of(mBitDepth);
    const uint32_t numFrames = (uint32_t)*ioNumBytes / (theInputFormat.mBytesPerPacket ? theInputFormat.mBytesPerPacket : bpf);
    if (numFrames > mFrameSize) return kALAC_ParamError;
    uint32_t nb = 0;
    uint64_t total = 0;
    std::vector<uint8_t> tmp((size_t)mFrameSize * bpf + 64);
    int32_t rc = EncodeBatch(theReadBuffer, numFrames, 0, tmp.data(), tmp.size(), &nb, &total);
    if (rc != ALAC_noErr) return rc;
    memcpy(theWriteBuffer, tmp.data(), total);
    *ioNumBytes = (int32_t)total;
    return ALAC_noErr;
}

// :1064-1073 (a no-op in the reference)
int32_t ALACEncoder::Finish() { return ALAC_noErr; }
