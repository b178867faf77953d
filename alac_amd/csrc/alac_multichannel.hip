// alac_multichannel.hip — streams of 3..8 channels.
//
// A packet of such a stream is a sequence of mono (ID_SCE) and stereo (ID_CPE) elements in the order of
// sChannelMaps (codec/ALACEncoder.cu:97-107), each element coded exactly like the one element of a mono / stereo
// packet from its own channels of the interleaved frame and its own coefficient rows, joined at BIT granularity,
// then one ID_END and the byte alignment (codec/ALACEncoder.cu:1034-1039; the element loop itself is Apple's, the
// fork dropped it and kept only the table and the decoder's loop codec/ALACDecoder.cu:600-990).
//
// So the encoder here is the mono / stereo pipeline run once per element over a gathered copy of its channels,
// followed by a splice: every element's bits are its one-element packet minus the trailing ID_END + padding (found
// from the last set bit: ID_END is '111' and the padding is zeros), with the 4-bit instance tag rewritten to the
// per-type running count.
#include "alac_dev.hpp"
#include "alac_kernels.hpp"

namespace alacdev {

uint32_t channel_elements(uint32_t numChannels, McElement *out)
{
    // sChannelMaps: 3 bits per channel index, ID_SCE = 0, ID_CPE = 1 (the table carries ID_SCE at the LFE positions)
    static const uint32_t maps[kMaxChannels] = {
        0, 1, (1u << 3) | 0, (0u << 9) | (1u << 3) | 0, (1u << 9) | (1u << 3) | 0, (0u << 15) | (1u << 9) | (1u << 3) | 0,
        (0u << 18) | (0u << 15) | (1u << 9) | (1u << 3) | 0, (0u << 21) | (1u << 15) | (1u << 9) | (1u << 3) | 0};
    if (numChannels < 1 || numChannels > kMaxChannels) return 0;
    uint32_t n = 0, mono = 0, stereo = 0;
    for (uint32_t ci = 0; ci < numChannels;) {
        const uint32_t tag = (maps[numChannels - 1] >> (ci * 3)) & 7u;
        out[n].first = ci;
        out[n].channels = tag == 1 ? 2 : 1;
        out[n].tag = (tag << 4) | (tag == 1 ? stereo++ : mono++);
        ci += out[n].channels;
        n++;
    }
    return n;
}

namespace {

__global__ __launch_bounds__(256) void k_mc_gather(const uint8_t *pcm, uint8_t *out, const uint32_t *numSamples,
                                                   uint64_t totalFrames, uint32_t frameSize, uint32_t numChannels,
                                                   uint32_t first, uint32_t channels, uint32_t bps)
{
    const uint64_t f = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (f >= totalFrames) return;
    const uint32_t p = (uint32_t)(f / frameSize), j = (uint32_t)(f % frameSize);
    if (numSamples && j >= numSamples[p]) return;
    const uint8_t *src = pcm + (f * numChannels + first) * bps;
    uint8_t *dst = out + f * channels * bps;
    const uint32_t n = channels * bps;
    for (uint32_t b = 0; b < n; b++) dst[b] = src[b];
}

__global__ __launch_bounds__(256) void k_mc_tables(const uint32_t *numSamples, uint32_t numPackets, const uint32_t *segFirst,
                                                   uint32_t numSegments, uint32_t count, uint32_t *nsOut, uint32_t *segOut)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (numSamples && i < (uint64_t)count * numPackets) nsOut[i] = numSamples[i % numPackets];
    if (segFirst && i <= (uint64_t)count * numSegments)
        segOut[i] = i == (uint64_t)count * numSegments
                        ? count * numPackets
                        : (uint32_t)(i / numSegments) * numPackets + segFirst[i % numSegments];
}

__global__ __launch_bounds__(256) void k_mc_sizes(McSpliceArgs A)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= A.numPackets) return;
    uint32_t total = 3;  // ID_END
    for (uint32_t e = 0; e < A.numElements; e++) {
        const uint64_t o0 = A.srcOffsets[e][p], o1 = A.srcOffsets[e][p + 1];
        uint32_t bits = 0;
        if (o1 > o0) {
            const uint32_t last = A.src[e][o1 - 1];
            // the packet ends ... 1 1 1 0*: the last set bit closes ID_END
            if (last) bits = (uint32_t)(o1 - o0 - 1) * 8u + (8u - (uint32_t)(__ffs((int)last) - 1)) - 3u;
        }
        A.elemBits[(uint64_t)e * A.numPackets + p] = bits;
        total += bits;
    }
    A.packetBytes[p] = (total + 7) / 8;
}

// n <= 32 bits of a byte string starting at bit b (MSB first)
__device__ __forceinline__ uint32_t bytes_fetch(const uint8_t *base, uint64_t b, uint32_t n)
{
    const uint8_t *q = base + (b >> 3);
    uint64_t w = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) w = (w << 8) | q[i];
    const uint32_t sh = (uint32_t)(b & 7);
    return (uint32_t)((w >> (40 - sh - n)) & (n >= 32 ? 0xffffffffull : ((1ull << n) - 1)));
}

__global__ __launch_bounds__(256) void k_mc_splice(McSpliceArgs A)
{
    // the packet as a list of bit segments: per element {7 constant bits: type + instance tag} {its bits from 7 on},
    // then {111}
    __shared__ uint32_t segStart[2 * kMaxChannels + 2], segLen[2 * kMaxChannels + 1], segVal[2 * kMaxChannels + 1];
    __shared__ int segSrc[2 * kMaxChannels + 1];
    __shared__ uint32_t numSegs;
    const uint32_t p = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t n = 0, pos = 0;
        for (uint32_t e = 0; e < A.numElements; e++) {
            const uint32_t bits = A.elemBits[(uint64_t)e * A.numPackets + p];
            if (bits < 7) continue;
            segStart[n] = pos, segLen[n] = 7, segVal[n] = A.el[e].tag, segSrc[n] = -1, pos += 7, n++;
            segStart[n] = pos, segLen[n] = bits - 7, segVal[n] = 7, segSrc[n] = (int)e, pos += bits - 7, n++;
        }
        segStart[n] = pos, segLen[n] = 3, segVal[n] = 7, segSrc[n] = -1, pos += 3, n++;
        segStart[n] = pos;
        numSegs = n;
    }
    __syncthreads();
    const uint32_t bytes = A.packetBytes[p];
    uint8_t *outp = A.out + A.offsets[p];
    const uint32_t ns = numSegs;
    for (uint32_t c = threadIdx.x; c * 4u < bytes; c += 256u) {
        const uint32_t b0 = c * 32u, b1 = b0 + 32u;
        uint32_t v = 0;
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t s0 = segStart[s], s1 = s0 + segLen[s];
            const uint32_t lo = s0 > b0 ? s0 : b0, hi = s1 < b1 ? s1 : b1;
            if (lo >= hi) continue;
            const uint32_t n = hi - lo;
            uint32_t bits;
            if (segSrc[s] < 0) {
                bits = (segVal[s] >> (s1 - hi)) & ((1u << n) - 1);  // constants are <= 7 bits
            } else {
                const int e = segSrc[s];
                bits = bytes_fetch(A.src[e] + A.srcOffsets[e][p], (uint64_t)segVal[s] + (lo - s0), n);
            }
            v |= n >= 32 ? bits : bits << (b1 - hi);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (c * 4u + k < bytes) outp[c * 4u + k] = (uint8_t)(v >> (24 - 8 * k));
    }
}

}  // namespace

void launch_mc_gather(const uint8_t *pcm, uint8_t *out, const uint32_t *numSamples, uint32_t numPackets,
                      uint32_t frameSize, uint32_t numChannels, uint32_t first, uint32_t channels, uint32_t bytesPerSample,
                      hipStream_t st)
{
    const uint64_t totalFrames = (uint64_t)numPackets * frameSize;
    hipLaunchKernelGGL(k_mc_gather, dim3((uint32_t)((totalFrames + 255) / 256)), dim3(256), 0, st, pcm, out, numSamples,
                       totalFrames, frameSize, numChannels, first, channels, bytesPerSample);
}

void launch_mc_tables(const uint32_t *numSamples, uint32_t numPackets, const uint32_t *segFirst, uint32_t numSegments,
                      uint32_t count, uint32_t *numSamplesOut, uint32_t *segFirstOut, hipStream_t st)
{
    if (!numSamples && !segFirst) return;
    const uint64_t n = (uint64_t)count * numPackets + 1;
    hipLaunchKernelGGL(k_mc_tables, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, numSamples, numPackets, segFirst,
                       numSegments, count, numSamplesOut, segFirstOut);
}

void launch_mc_splice(const McSpliceArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_mc_sizes, dim3((a.numPackets + 255) / 256), dim3(256), 0, st, a);
    launch_scan_sizes(a.packetBytes, a.offsets, a.numPackets, st);
    hipLaunchKernelGGL(k_mc_splice, dim3(a.numPackets), dim3(256), 0, st, a);
}

}  // namespace alacdev
