// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS library's access patterns (MI355X_MICROARCH.md, HBM section:
// "on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel below reads (or writes) a KNOWN number of distinct bytes exactly once from a buffer much larger than the 256 MiB
// Infinity Cache, in one of the shapes the encode / decode kernels use:
//   k_stream16        64 lanes x 16 B consecutive (k_pack's bulk copy, k_dec_stage)
//   k_rows192<OV>     the predictor staging (stage_load_fast, LPC = 1): a wave's 64 lanes read 16 B each, 12 consecutive lanes
//                     cover 192 contiguous bytes of ONE packet row, rows 16 KB apart (packet stride); OV = 1 re-reads 64 of every
//                     192 bytes one tile later (the history in front of a tile: ROWLEN 48 samples per TILE of 32)
//   k_rows128         the predictor staging since round 4 kept the history of a tile in LDS: 8 consecutive lanes cover the 128 new
//                     bytes of ONE packet row (16-bit stereo: one aligned line per row and tile), rows 16 KB apart, no re-reads;
//                     24-bit stereo stages 192 bytes per row and tile without overlap = k_rows192<0>
//   k_lane_rows16     one lane = one row: every lane streams through its own row with 16-B loads (entropy decoder's word
//                     stream, one-lane predictor of the decoder: 64 lanes, 64 different lines per instruction)
//   k_lane_rows4      the same with 4-byte loads (coder row loads of the planes are 256-B rows: covered by k_stream16)
//   k_store16 / k_store_lane16 / k_store_lane4   the matching store shapes
// Run:  hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o /tmp/fetch_cal
//       rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- /tmp/fetch_cal ; (again with --pmc WRITE_SIZE)
// and compare Counter_Value (KiB) per kernel with the bytes the program prints: tools/fetch_calibrate.py does that.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);  \
            return 1;                                                         \
        }                                                                     \
    } while (0)

typedef uint32_t U4 __attribute__((ext_vector_type(4)));

__global__ void k_stream16(const U4 *in, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const U4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// rows of rowBytes, `rows` of them; a wave takes 16 rows at a time?  No: like the staging, 64 lanes = 5.33 rows x 12 groups
template <int OV>
__global__ void k_rows192(const uint8_t *in, uint32_t rows, uint32_t rowBytes, uint32_t *sink)
{
    // wave w handles rows [32 w, 32 w + 32): per tile t, 6 rounds of 64 tasks (task = row q, group g of 12); a tile advances
    // 128 bytes (OV: and stages 192 from 64 bytes earlier); OV = 0: tiles are 192 bytes apart, no re-read
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t row0 = wave * 32;
    if (row0 >= rows) return;
    const uint32_t step = OV ? 128 : 192;
    uint32_t acc = 0;
    for (uint32_t off = 0; off + 192 <= rowBytes; off += step) {
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const uint32_t idx = it * 64 + lane, q = idx / 12, g = idx % 12;
            const U4 v = *(const U4 *)(in + (uint64_t)(row0 + q) * rowBytes + off + g * 16);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void k_rows128(const uint8_t *in, uint32_t rows, uint32_t rowBytes, uint32_t *sink)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t row0 = wave * 32;
    if (row0 >= rows) return;
    uint32_t acc = 0;
    for (uint32_t off = 0; off + 128 <= rowBytes; off += 128) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const uint32_t idx = it * 64 + lane, q = idx / 8, g = idx % 8;
            const U4 v = *(const U4 *)(in + (uint64_t)(row0 + q) * rowBytes + off + g * 16);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int BYTES>
__global__ void k_lane_rows(const uint8_t *in, uint32_t rows, uint32_t rowBytes, uint32_t *sink)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint8_t *p = in + (uint64_t)r * rowBytes;
    uint32_t acc = 0;
    for (uint32_t off = 0; off + BYTES <= rowBytes; off += BYTES) {
        if constexpr (BYTES == 16) {
            const U4 v = *(const U4 *)(p + off);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        } else {
            acc ^= *(const uint32_t *)(p + off);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void k_store16(U4 *out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const U4 v = {(uint32_t)i, 1, 2, 3};
        out[i] = v;
    }
}

template <int BYTES>
__global__ void k_store_lane(uint8_t *out, uint32_t rows, uint32_t rowBytes)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint8_t *p = out + (uint64_t)r * rowBytes;
    for (uint32_t off = 0; off + BYTES <= rowBytes; off += BYTES) {
        if constexpr (BYTES == 16) {
            const U4 v = {off, r, 2, 3};
            *(U4 *)(p + off) = v;
        } else {
            *(uint32_t *)(p + off) = off ^ r;
        }
    }
}

int main()
{
    const uint32_t rowBytes = 16384, rows = 131072;  // 2 GiB: 125 000-packet PCM scale, 8 x the Infinity Cache
    const uint64_t total = (uint64_t)rows * rowBytes;
    uint8_t *buf;
    uint32_t *sink;
    CK(hipMalloc((void **)&buf, total));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(buf, 1, total));
    CK(hipDeviceSynchronize());
    // expected distinct bytes per kernel
    const uint64_t tilesNo = rowBytes / 192, tilesOv = (rowBytes - 192) / 128 + 1;
    printf("EXPECT k_stream16 read %llu\n", (unsigned long long)total);
    printf("EXPECT k_rows192<0> read %llu\n", (unsigned long long)((uint64_t)rows * tilesNo * 192));
    printf("EXPECT k_rows192<1> read %llu requested %llu\n", (unsigned long long)((uint64_t)rows * (tilesOv * 128 + 64)),
           (unsigned long long)((uint64_t)rows * tilesOv * 192));
    printf("EXPECT k_rows128 read %llu\n", (unsigned long long)total);
    printf("EXPECT k_lane_rows<16> read %llu\n", (unsigned long long)total);
    printf("EXPECT k_lane_rows<4> read %llu\n", (unsigned long long)total);
    printf("EXPECT k_store16 write %llu\n", (unsigned long long)total);
    printf("EXPECT k_store_lane<16> write %llu\n", (unsigned long long)total);
    printf("EXPECT k_store_lane<4> write %llu\n", (unsigned long long)total);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_stream16, dim3(8192), dim3(256), 0, 0, (const U4 *)buf, total / 16, sink);
        hipLaunchKernelGGL(k_rows192<0>, dim3(rows / 32 / 4), dim3(256), 0, 0, buf, rows, rowBytes, sink);
        hipLaunchKernelGGL(k_rows192<1>, dim3(rows / 32 / 4), dim3(256), 0, 0, buf, rows, rowBytes, sink);
        hipLaunchKernelGGL(k_rows128, dim3(rows / 32 / 4), dim3(256), 0, 0, buf, rows, rowBytes, sink);
        hipLaunchKernelGGL(k_lane_rows<16>, dim3(rows / 256), dim3(256), 0, 0, buf, rows, rowBytes, sink);
        hipLaunchKernelGGL(k_lane_rows<4>, dim3(rows / 256), dim3(256), 0, 0, buf, rows, rowBytes, sink);
        hipLaunchKernelGGL(k_store16, dim3(8192), dim3(256), 0, 0, (U4 *)buf, total / 16);
        hipLaunchKernelGGL(k_store_lane<16>, dim3(rows / 256), dim3(256), 0, 0, buf, rows, rowBytes);
        hipLaunchKernelGGL(k_store_lane<4>, dim3(rows / 256), dim3(256), 0, 0, buf, rows, rowBytes);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
