/*
 * alac_synth.c — host build of the deterministic synthetic PCM generator (alac_synth_core.h).  Used by bench.py and
 * the tests to make the inputs both the HIP path and the CPU oracle encode; the device build of the same source is
 * alac_synth.hip (alac_hip_synth_pcm).
 */
#include "alac_synth_core.h"

void alac_synth_frame(uint64_t frameIndex, uint32_t numSamples, uint32_t bitDepth, uint32_t channels, uint8_t *out)
{
    alac_synth_frame_core(frameIndex, numSamples, bitDepth, channels, out);
}

/* numFrames consecutive frames starting at firstFrame, each frameSize sample-frames, back to back */
void alac_synth_pcm(uint64_t firstFrame, uint32_t numFrames, uint32_t frameSize, uint32_t bitDepth,
                    uint32_t channels, uint8_t *out)
{
    const size_t bps = bitDepth == 16 ? 2 : (bitDepth == 32 ? 4 : 3);
    const size_t frameBytes = (size_t)frameSize * channels * bps;
    for (uint32_t f = 0; f < numFrames; f++)
        alac_synth_frame(firstFrame + f, frameSize, bitDepth, channels, out + (size_t)f * frameBytes);
}
