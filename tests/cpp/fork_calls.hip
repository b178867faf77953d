// fork_calls — drives the drop-in ALACEncoder / ALACDecoder classes with EXACTLY the call sequence of the
// reference's convert utility (convert-utility/main.cu:411-414, 424-426, 462-548, 553-601 encode;
// :707-742 decode): device buffer of all packets -> InitializeSampling -> Encode(index) per packet, and
// Decode(..., index) per packet -> fillWriteBuffer into a device buffer.  Test harness only.
//
//   fork_calls <bits> <channels> <rate> <pcm in> <stream out> <sizes out> <pcm back out>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ALACAudioTypes.h"
#include "ALACDecoder.h"
#include "ALACEncoder.h"

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc != 8 && argc != 9) return 2;
    const bool fast = argc == 9 && std::string(argv[8]) == "fast";  // SetFastMode(true) before InitializeEncoder
    const uint32_t bits = atoi(argv[1]), ch = atoi(argv[2]), rate = atoi(argv[3]);
    const std::vector<uint8_t> pcm = slurp(argv[4]);
    const uint32_t frames = kALACDefaultFramesPerPacket;

    AudioFormatDescription in, out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    in.mFormatID = kALACFormatLinearPCM;
    in.mSampleRate = rate;
    in.mChannelsPerFrame = ch;
    in.mBitsPerChannel = bits;
    in.mFormatFlags = kALACFormatFlagIsSignedInteger | kALACFormatFlagIsPacked;
    in.mBytesPerPacket = in.mBytesPerFrame = (bits >> 3) * ch;
    in.mFramesPerPacket = 1;
    out.mFormatID = kALACFormatAppleLossless;
    out.mSampleRate = rate;
    out.mFormatFlags = bits == 16 ? 1 : bits == 20 ? 2 : bits == 24 ? 3 : 4;
    out.mFramesPerPacket = frames;
    out.mChannelsPerFrame = ch;

    // ---- EncodeALAC's sequence ----
    const int32_t inPacketBytes = ch * (bits >> 3) * frames;
    const int32_t outPacketBytes = inPacketBytes + kALACMaxEscapeHeaderBytes;
    const int X = (int)(pcm.size() / inPacketBytes) + 1;
    ALACEncoder *enc = new ALACEncoder;
    enc->SetFrameSize(frames);
    enc->SetFastMode(fast);
    if (enc->InitializeEncoder(out, X) != ALAC_noErr) return 3;
    uint32_t cookieSize = enc->GetMagicCookieSize(ch);
    std::vector<uint8_t> cookie(cookieSize);
    enc->GetMagicCookie(cookie.data(), &cookieSize);

    std::vector<int32_t> outBytes(X, 0);
    void *dIn = nullptr;
    if (hipMalloc(&dIn, (size_t)X * inPacketBytes) != hipSuccess) return 4;
    int index = 0;
    size_t remaining = pcm.size(), src = 0;
    while ((size_t)inPacketBytes <= remaining) {
        outBytes[index] = inPacketBytes;
        (void)hipMemcpy((uint8_t *)dIn + (size_t)index * inPacketBytes, pcm.data() + src, inPacketBytes, hipMemcpyHostToDevice);
        src += inPacketBytes;
        remaining -= inPacketBytes;
        index++;
    }
    if (remaining) {
        outBytes[index] = (int32_t)remaining;
        (void)hipMemcpy((uint8_t *)dIn + (size_t)index * inPacketBytes, pcm.data() + src, remaining, hipMemcpyHostToDevice);
        index++;
    }
    const int numPackets = index;
    enc->InitializeSampling(dIn, in, X, outBytes.data());
    if (enc->LastStatus() != ALAC_noErr) return 5;

    std::vector<uint8_t> readBuf(inPacketBytes), writeBuf(outPacketBytes), stream;
    std::vector<uint32_t> sizes;
    for (int i = 0; i < numPackets; i++) {
        int32_t numBytes = outBytes[i];
        if (enc->Encode(in, in, readBuf.data(), writeBuf.data(), &numBytes, i) != ALAC_noErr) return 6;
        stream.insert(stream.end(), writeBuf.begin(), writeBuf.begin() + numBytes);
        sizes.push_back((uint32_t)numBytes);
    }
    delete enc;
    (void)hipFree(dIn);
    FILE *f = fopen(argv[5], "wb");
    fwrite(stream.data(), 1, stream.size(), f);
    fclose(f);
    f = fopen(argv[6], "wb");
    fwrite(sizes.data(), 4, sizes.size(), f);
    fclose(f);

    // ---- DecodeALAC's sequence ----
    ALACDecoder *dec = new ALACDecoder;
    if (dec->Init(cookie.data(), cookieSize, numPackets) != ALAC_noErr) return 7;
    const uint32_t bytesPerFrame = ch * (bits >> 3);
    std::vector<uint8_t> pktBuf(outPacketBytes);
    BitBuffer bb;
    std::vector<int32_t> hNumBytes(numPackets, 0);
    size_t off = 0;
    for (int i = 0; i < numPackets; i++) {
        memcpy(pktBuf.data(), stream.data() + off, sizes[i]);
        BitBufferInit(&bb, pktBuf.data(), sizes[i]);
        uint32_t numFrames = 0;
        if (dec->Decode(&bb, frames, ch, &numFrames, bytesPerFrame, i) != ALAC_noErr) return 8;
        hNumBytes[i] = (int32_t)(numFrames * bytesPerFrame);
        off += sizes[i];
    }
    uint8_t *dOut = nullptr;
    if (hipMalloc((void **)&dOut, (size_t)numPackets * inPacketBytes) != hipSuccess) return 9;
    dec->fillWriteBuffer(dOut, ch, inPacketBytes, numPackets);
    if (dec->LastStatus() != ALAC_noErr) return 10;
    f = fopen(argv[7], "wb");
    std::vector<uint8_t> back(inPacketBytes);
    for (int i = 0; i < numPackets; i++) {
        (void)hipMemcpy(back.data(), dOut + (size_t)i * inPacketBytes, hNumBytes[i], hipMemcpyDeviceToHost);
        fwrite(back.data(), 1, hNumBytes[i], f);
    }
    fclose(f);
    delete dec;
    (void)hipFree(dOut);
    printf("packets %d bytes %zu\n", numPackets, stream.size());
    return 0;
}
