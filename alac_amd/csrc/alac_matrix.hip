// alac_matrix.hip — the stereo-matrixing stage surface of the reference (codec/matrixlib.h:41-60) on the GPU:
// mix16 / mix20 / mix24 / mix32 / copy20ToPredictor with the reference prototypes, one thread per sample-frame,
// coalesced along the sample index (an HBM-bound element-wise pass; the encode hot path does not call these — it fuses
// the same arithmetic into the predictor's staging, alac_encode_v1.hip stage_store).
//
// Math: codec/matrix_enc.cu:72-118 (16-bit), :120-183 (20-bit), :186-323 (24-bit), :330-425 (32-bit), :428-445
// (copy20ToPredictor).  Buffers may be device pointers (the fork's convention: its mixNN launch kernels on what they
// are given) or host pointers (Apple's upstream convention); host buffers are staged through the device.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "alac/matrixlib.h"

namespace {

// sample `idx` (channel-interleaved element index) of a packed little-endian PCM buffer, sign extended
template <int DEPTH>
__device__ __forceinline__ int32_t pcm_at(const uint8_t *in, uint64_t idx)
{
    if constexpr (DEPTH == 16) {
        return ((const int16_t *)in)[idx];
    } else if constexpr (DEPTH == 32) {
        return ((const int32_t *)in)[idx];
    } else {
        const uint8_t *p = in + 3 * idx;
        const uint32_t w = ((uint32_t)p[2] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 8);
        return (int32_t)w >> (DEPTH == 20 ? 12 : 8);  // 20-bit: left-justified in 3 bytes, right-aligned out
    }
}

template <int DEPTH>
__global__ void k_mix(const uint8_t *in, uint32_t stride, int32_t *u, int32_t *v, uint32_t n, int32_t mixbits,
                      int32_t mixres, uint16_t *shiftUV, int32_t shift, bool writeShift)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int32_t l = pcm_at<DEPTH>(in, (uint64_t)j * stride), r = pcm_at<DEPTH>(in, (uint64_t)j * stride + 1);
    if (writeShift) {
        const uint32_t mask = (1u << shift) - 1u;
        shiftUV[2 * j] = (uint16_t)((uint32_t)l & mask);
        shiftUV[2 * j + 1] = (uint16_t)((uint32_t)r & mask);
    }
    l >>= shift;
    r >>= shift;
    if (mixres != 0) {
        const int32_t m2 = (1 << mixbits) - mixres;
        u[j] = (mixres * l + m2 * r) >> mixbits;
        v[j] = l - r;
    } else {
        u[j] = l;
        v[j] = r;
    }
}

__global__ void k_copy20(const uint8_t *in, uint32_t stride, int32_t *out, uint32_t n)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = pcm_at<20>(in, (uint64_t)j * stride);
}

bool on_device(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

struct Staged {  // device twin of a host buffer
    void *d = nullptr;
    void *h = nullptr;
    size_t bytes = 0;
    bool own = false;
    bool ok = true;
    Staged(void *p, size_t n, bool dev, bool copyIn) : h(p), bytes(n)
    {
        if (!p || n == 0) return;
        if (dev) {
            d = p;
            return;
        }
        own = true;
        ok = hipMalloc(&d, n) == hipSuccess && (!copyIn || hipMemcpy(d, p, n, hipMemcpyHostToDevice) == hipSuccess);
    }
    void back() const
    {
        if (own && ok && d) (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
    }
    ~Staged()
    {
        if (own && d) (void)hipFree(d);
    }
};

template <int DEPTH>
void mix_any(void *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres,
             uint16_t *shiftUV, int32_t bytesShifted, bool wantShift)
{
    if (!in || !u || !v || numSamples <= 0 || stride < 2) return;
    constexpr size_t BPS = DEPTH == 16 ? 2 : (DEPTH == 32 ? 4 : 3);
    const size_t n = (size_t)numSamples;
    const bool dev = on_device(in);
    const bool ws = wantShift && shiftUV != nullptr;
    Staged sIn(in, ((n - 1) * stride + 2) * BPS, dev, true), sU(u, n * 4, dev, false), sV(v, n * 4, dev, false),
        sS(ws ? shiftUV : nullptr, n * 4, dev, false);
    if (!sIn.ok || !sU.ok || !sV.ok || !sS.ok) return;
    hipLaunchKernelGGL(k_mix<DEPTH>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, (const uint8_t *)sIn.d, stride,
                       (int32_t *)sU.d, (int32_t *)sV.d, (uint32_t)n, mixbits, mixres, (uint16_t *)sS.d, bytesShifted * 8, ws);
    if (!dev) {
        (void)hipDeviceSynchronize();
        sU.back();
        sV.back();
        sS.back();
    }
}

}  // namespace

extern "C" {

void mix16(int16_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres)
{
    mix_any<16>(in, stride, u, v, numSamples, mixbits, mixres, nullptr, 0, false);
}

void mix20(uint8_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres)
{
    mix_any<20>(in, stride, u, v, numSamples, mixbits, mixres, nullptr, 0, false);
}

// the shifted-off bytes are produced only when bytesShifted != 0 (codec/matrix_enc.cu:293-323)
void mix24(uint8_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres,
           uint16_t *shiftUV, int32_t bytesShifted)
{
    if (bytesShifted < 0 || bytesShifted > 2) return;
    mix_any<24>(in, stride, u, v, numSamples, mixbits, mixres, shiftUV, bytesShifted, bytesShifted != 0);
}

// matrixed 32-bit input always goes through the shift buffer, plain de-interleaving only when bytesShifted != 0
// (codec/matrix_enc.cu:393-425)
void mix32(int32_t *in, uint32_t stride, int32_t *u, int32_t *v, int32_t numSamples, int32_t mixbits, int32_t mixres,
           uint16_t *shiftUV, int32_t bytesShifted)
{
    if (bytesShifted < 0 || bytesShifted > 2) return;
    mix_any<32>(in, stride, u, v, numSamples, mixbits, mixres, shiftUV, bytesShifted, mixres != 0 || bytesShifted != 0);
}

void copy20ToPredictor(uint8_t *in, uint32_t stride, int32_t *out, int32_t numSamples)
{
    if (!in || !out || numSamples <= 0 || stride < 1) return;
    const size_t n = (size_t)numSamples;
    const bool dev = on_device(in);
    Staged sIn(in, ((n - 1) * stride + 1) * 3, dev, true), sOut(out, n * 4, dev, false);
    if (!sIn.ok || !sOut.ok) return;
    hipLaunchKernelGGL(k_copy20, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, (const uint8_t *)sIn.d, stride,
                       (int32_t *)sOut.d, (uint32_t)n);
    if (!dev) {
        (void)hipDeviceSynchronize();
        sOut.back();
    }
}

}  // extern "C"
