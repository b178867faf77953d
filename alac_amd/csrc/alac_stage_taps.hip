// alac_stage_taps.hip — pc_block for ANY tap count (the general loop of codec/dp_enc.c:341-387, the "deep LPC"
// shapes: numactive 9..30), tap-parallel: one chain per 32-lane half of a wave, lane k owns tap k (coefficient a_k
// and the sample in[j-1-k] in registers), the tap sum is a wave shuffle reduction and the early-exit coefficient
// walk becomes a suffix scan across the lanes.
//
// Per step (one residual):
//   history   x_k <- x_{k-1} with ONE DPP wave_shr:1 (lane 0 of each half takes the new sample)
//   top       broadcast of lane `na`'s x (in[j-na-1])
//   sum       32-lane butterfly: quad_perm xor1, xor2, row_half_mirror, row_mirror (DPP) + one cross-row shuffle
//   update    tap k is touched iff |del| > S_k = sum_{i>k} (na-i) t_i: inclusive suffix sums by row_shl:1,2,4,8
//             (DPP, zeros shifted in) + the upper row's total for the lower row; a_k -= sign(del) sign(dd_k)
// Samples and residuals move 32 at a time (lane s of a half loads in[jb+s] / stores pc[jb+s], coalesced); the
// per-step value is picked with a shuffle.
//
// The encoder's own passes never come here (they use 4 or 8 taps on the 2-lanes-per-chain mapping of
// alac_lms.hpp, which is faster for <= 8 taps); this is the stage-level entry point alac_hip_pc_block for the
// tap counts a general ALAC encoder may ask for.  Range: chanbits <= 24 and denshift >= 5 keep the threshold sums
// inside int32 exactly like the reference's running del0; other shapes stay on the lane-serial kernel.
#include "alac_dev.hpp"
#include "alac_kernels.hpp"

namespace alacdev {

namespace {

constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140, kWaveShr1 = 0x138;

template <int CTRL>
__device__ __forceinline__ int32_t dpp0(int32_t v)  // out-of-range source lanes read 0
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// sum over the 32 lanes of each half, result in every lane
__device__ __forceinline__ int32_t half_sum(int32_t v)
{
    v += dpp0<kQuadXor1>(v);
    v += dpp0<kQuadXor2>(v);
    v += dpp0<kRowHalfMirror>(v);
    v += dpp0<kRowMirror>(v);
    v += __shfl_xor(v, 16, 32);
    return v;
}

// exclusive suffix sum over the 32 lanes of each half: S_k = sum_{i>k} w_i
__device__ __forceinline__ int32_t half_suffix_exclusive(int32_t w, int k)
{
    int32_t v = w;
    v += dpp0<0x101>(v);  // row_shl:1  lane k takes lane k+1 (0 past the end of the 16-lane row)
    v += dpp0<0x102>(v);  // row_shl:2
    v += dpp0<0x104>(v);
    v += dpp0<0x108>(v);
    const int32_t upper = __shfl(v, 16, 32);  // total of lanes 16..31 of this half
    v += k < 16 ? upper : 0;
    return v - w;
}

}  // namespace

__global__ __launch_bounds__(64) void k_pc_block_taps(const int32_t *in, int32_t *pc, uint32_t rows, uint32_t stride,
                                                      int32_t num, int16_t *coefs, int32_t na, uint32_t chanbits,
                                                      uint32_t denshift)
{
    const int lane = threadIdx.x, k = lane & 31, half = lane >> 5;
    const uint32_t r = blockIdx.x * 2u + half;
    const bool row = r < rows;
    const int32_t *src = in + (uint64_t)(row ? r : 0) * stride;
    int32_t *dst = pc + (uint64_t)(row ? r : 0) * stride;
    const uint32_t chanshift = 32 - chanbits;
    const int32_t denhalf = denshift ? (1 << (denshift - 1)) : 0;
    const bool tap = k < na;
    int32_t a = (row && tap) ? (int32_t)coefs[(uint64_t)r * 32 + k] : 0;

    // warm-up positions (dp_enc.c:90, :108-112): lane t writes pc[t], t = 0 .. na
    if (row && k <= na) dst[k] = k == 0 ? src[0] : sext(src[k] - src[k - 1], chanshift);

    // history: x_k = in[j - 1 - k] at j = na + 1; lane na then holds top = in[0]
    int32_t x = (row && k <= na) ? src[na - k] : 0;
    const int32_t wgt = tap ? na - k : 0;
    const int32_t round = (1 << denshift) - 1;

    for (int32_t jb = na + 1; jb < num; jb += 32) {
        // the block's 32 samples, one per lane (coalesced); stride covers max(num, na + 1) readable samples
        int32_t nxt = (row && jb + k < num) ? src[jb + k] : 0;
        int32_t outv = 0;
        const int steps = min(32, num - jb);
        for (int s = 0; s < steps; s++) {
            const int32_t cur = __shfl(nxt, s, 32);
            const int32_t top = __shfl(x, na, 32);
            const int32_t dd = tap ? top - x : 0;
            const int32_t sum1 = half_sum(-a * dd);  // sum a_k (pin[-k] - top), int32 wrap as in the reference
            const int32_t del = sext(cur - top - ((sum1 + denhalf) >> denshift), chanshift);
            outv = k == s ? del : outv;
            // coefficient walk (dp_enc.c:365-385)
            const int32_t sg = (del > 0) - (del < 0);
            const int32_t ab = dd < 0 ? -dd : dd;
            const int32_t t = (sg > 0 ? ab : ab + round) >> denshift;  // (sgn dd) >> ds resp. -((-sgn dd) >> ds)
            const int32_t S = half_suffix_exclusive(wgt * t, k);
            const int32_t adel = del < 0 ? -del : del;
            const int32_t sd = (dd > 0) - (dd < 0);
            if (tap && sg != 0 && adel > S) a = (int16_t)(a - sg * sd);
            // slide the window: lane k takes lane k-1's sample, lane 0 of each half the new one
            const int32_t shifted = __builtin_amdgcn_update_dpp(0, x, kWaveShr1, 0xf, 0xf, false);
            x = k == 0 ? cur : shifted;
        }
        if (row && jb + k < num) dst[jb + k] = outv;
    }
    if (row && tap) coefs[(uint64_t)r * 32 + k] = (int16_t)a;
}

bool pc_block_taps_ok(int32_t num, int32_t na, uint32_t chanbits, uint32_t denshift)
{
    return na >= 1 && na <= 30 && chanbits >= 1 && chanbits <= 24 && denshift >= 5 && denshift <= 15 && num >= 0;
}

void launch_pc_block_taps(const int32_t *in, int32_t *pc, uint32_t rows, uint32_t stride, int32_t num, int16_t *coefs,
                          int32_t na, uint32_t chanbits, uint32_t denshift, hipStream_t st)
{
    hipLaunchKernelGGL(k_pc_block_taps, dim3((rows + 1) / 2), dim3(64), 0, st, in, pc, rows, stride, num, coefs, na,
                       chanbits, denshift);
}

}  // namespace alacdev
