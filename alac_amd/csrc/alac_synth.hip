// alac_synth.hip — the synthetic PCM generator on the device (BASELINE.json configs[3] / SURVEY.md §8d config 4:
// "1 000 000 independent segments generated on device from the same recipe"): alac_hip_synth_pcm writes frames
// [first_frame, first_frame + num_frames) straight into HBM, no 2 GB host generation + H2D per rank.  Same source as
// the host generator (alac_synth_core.h), so the bytes are identical; pinned by tests against known_answers.json.
//
// A frame is a serial recurrence (RNG, AR(2), sine states) over its samples, frames are independent: one thread per
// frame.  A thread's stores walk its own frame, so consecutive lanes are a whole frame apart (uncoalesced); this is
// set-up, never timed: 125 000 frames take a few tens of ms.
#include <hip/hip_runtime.h>

#define ALAC_SYNTH_FN static __device__ __forceinline__
#include "alac_synth_core.h"
#include "alac_hip.h"

namespace {

__global__ __launch_bounds__(64) void k_synth(uint64_t firstFrame, uint32_t numFrames, uint32_t frameSize, uint32_t bitDepth,
                                              uint32_t channels, uint8_t *out)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= numFrames) return;
    const uint64_t bps = bitDepth == 16 ? 2 : (bitDepth == 32 ? 4 : 3);
    alac_synth_frame_core(firstFrame + f, frameSize, bitDepth, channels, out + (uint64_t)f * frameSize * channels * bps);
}

}  // namespace

extern "C" int32_t alac_hip_synth_pcm(alac_hip_ctx *ctx, uint64_t first_frame, uint32_t num_frames,
                                      const alac_hip_format *fmt, uint8_t *d_out)
{
    if (!ctx || !fmt || !d_out) return ALAC_HIP_ParamError;
    if (!(fmt->bit_depth == 16 || fmt->bit_depth == 20 || fmt->bit_depth == 24 || fmt->bit_depth == 32)) return ALAC_HIP_ParamError;
    if (fmt->num_channels < 1 || fmt->num_channels > 2 || fmt->frame_size == 0) return ALAC_HIP_ParamError;
    if (num_frames == 0) return ALAC_HIP_noErr;
    hipLaunchKernelGGL(k_synth, dim3((num_frames + 63) / 64), dim3(64), 0, (hipStream_t)alac_hip_stream(ctx), first_frame,
                       num_frames, fmt->frame_size, fmt->bit_depth, fmt->num_channels, d_out);
    return hipGetLastError() == hipSuccess ? ALAC_HIP_noErr : ALAC_HIP_ParamError;
}
