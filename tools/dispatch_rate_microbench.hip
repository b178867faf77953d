// Workgroup dispatch rate of an MI355X: how long a launch of G workgroups of T threads takes when the kernel does
// (almost) nothing.  The packer (k_pack, one workgroup per packet) was suspected to be bound by this.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate_microbench.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_empty(uint32_t *out)
{
    if (out == nullptr && threadIdx.x == 12345) out[0] = 1;
}
// one dependent global load per workgroup (what a packet record costs), then exit
__global__ void k_one_load(const uint32_t *in, uint32_t *out)
{
    const uint32_t v = in[blockIdx.x * 16];
    if (v == 0xdeadbeefu) out[0] = v;
}
// load -> dependent load -> store: the shape of a one-round-trip copy of `bytes` per workgroup
__global__ void k_copy(const uint4 *in, uint4 *out, uint32_t vecPerWg)
{
    const uint4 *src = in + (uint64_t)blockIdx.x * vecPerWg;
    uint4 *dst = out + (uint64_t)blockIdx.x * vecPerWg;
    for (uint32_t i = threadIdx.x; i < vecPerWg; i += blockDim.x) dst[i] = src[i];
}

// the packer's shape: two source strings per workgroup in sparse slots (slotVec uint4 apart), funnel-shifted by `sh`
// bits, five dword loads per 16-byte store (DWORDS) or one 16-byte load (sh ignored)
template <bool DWORDS>
__global__ void k_copy_sparse(const uint32_t *in, uint4 *out, uint32_t vecPerWg, uint32_t slotWords, uint32_t sh)
{
    const uint32_t *src = in + (uint64_t)blockIdx.x * slotWords;
    uint4 *dst = out + (uint64_t)blockIdx.x * vecPerWg;
    for (uint32_t i = threadIdx.x; i < vecPerWg; i += blockDim.x) {
        if constexpr (DWORDS) {
            uint32_t a[5];
#pragma unroll
            for (int q = 0; q < 5; q++) a[q] = src[4 * i + q + 1];
            uint4 v;
            v.x = __builtin_bswap32(__builtin_amdgcn_alignbit(a[0], a[1], 32 - sh));
            v.y = __builtin_bswap32(__builtin_amdgcn_alignbit(a[1], a[2], 32 - sh));
            v.z = __builtin_bswap32(__builtin_amdgcn_alignbit(a[2], a[3], 32 - sh));
            v.w = __builtin_bswap32(__builtin_amdgcn_alignbit(a[3], a[4], 32 - sh));
            dst[i] = v;
        } else {
            dst[i] = ((const uint4 *)src)[i];
        }
    }
}

template <class F>
static float time_ms(F &&launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const uint32_t G = 125000;
    uint32_t *in, *out;
    hipMalloc(&in, (size_t)G * 65664 + 64);
    hipMalloc(&out, (size_t)G * 8192);
    hipMemset(in, 0, (size_t)G * 65664);
    for (uint32_t g : {10000u, 125000u}) {
        for (uint32_t t : {64u, 128u, 256u, 512u, 1024u}) {
            const float e = time_ms([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(t), 0, 0, out); }, 10);
            const float l = time_ms([&] { hipLaunchKernelGGL(k_one_load, dim3(g), dim3(t), 0, 0, in, out); }, 10);
            const float c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(t), 0, 0, (const uint4 *)in, (uint4 *)out, 6800u / 16); }, 10);
            printf("workgroups %6u x %4u threads: empty %.4f ms (%.1f WG/us)  one-load %.4f ms  copy-6.8KB %.4f ms (%.2f TB/s r+w)\n", g, t, e,
                   g / (e * 1e3), l, c, 2.0 * g * 6800.0 / (c * 1e-3) / 1e12);
        }
    }
    for (uint32_t slot : {6800u / 4, 4096u, 8208u, 16416u}) {
        const float c4 = time_ms([&] { hipLaunchKernelGGL(k_copy_sparse<false>, dim3(G), dim3(256), 0, 0, (const uint32_t *)in, (uint4 *)out, 6800u / 16, slot, 7u); }, 10);
        const float c1 = time_ms([&] { hipLaunchKernelGGL(k_copy_sparse<true>, dim3(G), dim3(256), 0, 0, (const uint32_t *)in, (uint4 *)out, 6800u / 16, slot, 7u); }, 10);
        printf("125000 workgroups, 6.8 KB each out of a %u-byte slot: 16-byte loads %.4f ms, five dword loads + alignbit %.4f ms\n", slot * 4, c4, c1);
    }
    // the same copy from a persistent grid
    for (uint32_t wgs : {2048u, 4096u, 8192u}) {
        const float c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, (const uint4 *)in, (uint4 *)out, (uint32_t)((uint64_t)G * 6800 / 16 / wgs)); }, 10);
        printf("persistent %5u x 256 threads, contiguous slabs: copy %.4f ms (%.2f TB/s r+w)\n", wgs, c, 2.0 * G * 6800.0 / (c * 1e-3) / 1e12);
    }
    return 0;
}
