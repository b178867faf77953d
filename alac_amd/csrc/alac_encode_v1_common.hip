// alac_encode_v1_common.hip — the kernels of the tap-parallel encode pipeline that do not depend on the bit depth, compiled
// once (alac_encode_v1_impl.hpp declares the v1c_* launch wrappers the per-depth translation units call).
#define ALAC_V1_COMMON_TU 1
#include "alac_encode_v1_impl.hpp"

namespace alacdev {

void v1c_decide_fast(uint32_t nseg, hipStream_t st, const V1Args &A)
{
    hipLaunchKernelGGL(k_decide_fast, dim3((nseg + 255) / 256), dim3(256), 0, st, A);
}

void v1c_gol_count1(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits)
{
    if (ch == 2) hipLaunchKernelGGL(k_gol_count1<2>, dim3(cblocks, 5), dim3(64), 0, st, A, chanBits);
    else hipLaunchKernelGGL(k_gol_count1<1>, dim3(cblocks, 5), dim3(64), 0, st, A, chanBits);
}

void v1c_gol_count2(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits)
{
    if (ch == 2) hipLaunchKernelGGL(k_gol_count2<2>, dim3(cblocks, 2), dim3(64), 0, st, A, chanBits);
    else hipLaunchKernelGGL(k_gol_count2<1>, dim3(cblocks, 2), dim3(64), 0, st, A, chanBits);
}

void v1c_gol_count2_w(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits)
{
    const dim3 grid((2 * cblocks + kWavesPerWg - 1) / kWavesPerWg), block(64 * kWavesPerWg);
    if (ch == 2) hipLaunchKernelGGL(k_gol_count2_w<2>, grid, block, 0, st, A, cblocks, chanBits);
    else hipLaunchKernelGGL(k_gol_count2_w<1>, grid, block, 0, st, A, cblocks, chanBits);
}

void v1c_class_layout(int ch, uint32_t nseg, hipStream_t st, const V1Args &A, uint32_t *blockCnt)
{
    const dim3 grid((nseg + 1023) / 1024), block(1024);
    if (ch == 2) {
        hipLaunchKernelGGL(k_class_count<2>, grid, block, 0, st, A, blockCnt);
        hipLaunchKernelGGL(k_class_assign<2>, grid, block, 0, st, A, blockCnt);
    } else {
        hipLaunchKernelGGL(k_class_count<1>, grid, block, 0, st, A, blockCnt);
        hipLaunchKernelGGL(k_class_assign<1>, grid, block, 0, st, A, blockCnt);
    }
}

void v1c_splice_split(int ch, uint32_t nseg, hipStream_t st, const V1Args &A)
{
    if (ch == 2) hipLaunchKernelGGL(k_splice_split<2>, dim3(nseg * 2), dim3(64), 0, st, A);
    else hipLaunchKernelGGL(k_splice_split<1>, dim3(nseg), dim3(64), 0, st, A);
}

void v1c_gol_final(int ch, uint32_t cblocks, hipStream_t st, const V1Args &A, uint32_t chanBits)
{
    if (ch == 2) hipLaunchKernelGGL(k_gol_final<2>, dim3(cblocks), dim3(64), 0, st, A, chanBits);
    else hipLaunchKernelGGL(k_gol_final<1>, dim3(cblocks), dim3(64), 0, st, A, chanBits);
}

}  // namespace alacdev
