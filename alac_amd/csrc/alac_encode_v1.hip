// alac_encode_v1.hip — dispatcher of the tap-parallel encode pipeline: regime selection, argument block, and the call
// into the per-depth launcher (alac_encode_v1_impl.hpp, instantiated in alac_encode_v1_d16/20/24/32.hip).
#include <cstdlib>
#include "alac_encode_v1_types.hpp"

namespace alacdev {

// init_coefs for every row of the working state (codec/ALACEncoder.cu:1524-1531)
__global__ void k_init_state(int16_t *state, uint32_t numSegments)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numSegments * 64u) return;
    const uint32_t k = i & 15;
    state[i] = (int16_t)(k == 0 ? 1216 : k == 1 ? -928 : k == 2 ? -64 : 0);
}

__global__ void k_check_segments(const uint32_t *segFirst, uint32_t numSegments, uint32_t numPackets, uint32_t maxSeg, uint32_t *err,
                                 uint32_t *segBad)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= numSegments) return;
    const uint32_t a = segFirst[s], b = segFirst[s + 1];
    const bool bad = b < a || b > numPackets || b - a > maxSeg || (s == 0 && a != 0) || (s + 1 == numSegments && b != numPackets);
    if (bad && err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (bad && segBad) *segBad = 1u;  // what the launches behind this one on the stream test (EncodeArgs::segBad)
}

void launch_check_segments(const uint32_t *segFirst, uint32_t numSegments, uint32_t numPackets, uint32_t maxSeg, uint32_t *err,
                           uint32_t *segBad, hipStream_t st)
{
    if (segBad) (void)hipMemsetAsync(segBad, 0, 4, st);
    hipLaunchKernelGGL(k_check_segments, dim3((numSegments + 255) / 256), dim3(256), 0, st, segFirst, numSegments, numPackets, maxSeg, err,
                       segBad);
}

// chains beyond what one 2-lane predictor wave per SIMD holds (1024 SIMDs x 32 chains x 2): throughput regime
bool v1_throughput_regime(uint32_t numSegments, uint32_t channels, const AlacOptions &opt)
{
    if (opt.thru >= 0) return opt.thru != 0;
    return (uint64_t)numSegments * (channels > 2 ? 2 : channels) > 65536;
}

// Four lanes per chain ("tiny") or two ("latency") below the throughput regime.  Every launch of these regimes is made of
// single-wave workers that want a SIMD to themselves (1024 SIMDs): five per 64 chains with four lanes per chain, three with two.
// The four-lane form has the shorter serial chain and wins while ITS workers fit (measured crossover 5 500-6 000 packets = 11 000-
// 12 000 chains, 16- and 24-bit); then the two-lane form while its workers fit (21 845 chains: at 10 000 packets 938 workers);
// behind that cliff — 10 500 -> 11 000 packets: 1.46 -> 2.26 ms, some SIMDs now carry two predictor waves — the four-lane form,
// already past its own cliff and flat, is faster again until ~17 000 packets (11 000 / 12 000 / 14 000 / 16 000 / 18 000 packets,
// four against two lanes: 1.81 / 2.26, 1.82 / 2.44, 2.23 / 2.49, 2.34 / 2.54, 2.64 / 2.61 ms; profiles/r04/encode_regime_sweep.log).
// Until round 4 the rule was "four lanes up to 4096 chains".  Mono streams (no mixRes search, whose five passes are where the
// four-lane form gains most) cross over earlier at both ends: 10 000 / 11 000 chains 0.93 / 0.99 against 1.07 / 1.00 ms, and
// 26 000 / 28 000 chains 1.55 / 1.77 against 1.97 / 1.78 — there the four-lane workers reach two per SIMD (26 214 chains).
bool v1_narrow_regime(uint64_t chains, uint32_t channels, const AlacOptions &opt)
{
    if (opt.narrow >= 0) return opt.narrow != 0;
    const bool mono = channels == 1;
    if (chains <= (mono ? 10240u : 11264u)) return true;
    if (chains <= 21760) return false;  // 3 workers per 64 chains <= 1020
    return chains <= (mono ? 26112u : 34816u);
}

hipError_t launch_encode_v1(uint32_t depth, uint32_t channels, const EncodeArgs &ea, const PackArgs &pa,
                            const V1Buffers &vb, const V1Streams &vs, uint32_t numPackets, uint32_t maxSegPackets,
                            hipStream_t st, hipEvent_t *ev)
{
    V1Args A;
    A.S.pcm = ea.pcm;
    A.S.numSamples = ea.numSamples;
    A.S.segFirst = ea.segFirst;
    A.S.numSegments = ea.numSegments;
    A.S.frameSize = ea.frameSize;
    A.S.pos = 0;
    A.S.segBegin = 0;
    A.S.segEnd = ea.numSegments;
    A.S.numPackets = ea.numPackets;
    A.S.segMax = ea.segMax;
    A.state = vb.state;
    A.recs = ea.recs;
    A.resA = vb.resA;
    A.resB = vb.resB;
    A.resC = vb.resC;
    A.bits1 = vb.bits1;
    A.cost2 = vb.cost2;
    A.chainsPad = vb.chainsPad;
    A.bitWords = ea.bitWords;
    A.wcap = ea.wcap;
    A.dumpSlot = numPackets * 2;
    {
        // Latency regime (about one wave per SIMD: up to ~2 x 1024 x 32 chains): idle lanes should not slow their
        // wave down (idleFast).  With many waves per SIMD the machine is throughput bound and the extra work of idle lanes
        // costs more than the checked paths (measured: 125 000 packets 18.6 ms vs 20.3 ms).
        const uint64_t chains = (uint64_t)ea.numSegments * channels;
        A.thru = v1_throughput_regime(ea.numSegments, channels, vb.opt) ? 1u : 0u;
        A.idleFast = A.thru ? 0u : 1u;
        A.narrow = v1_narrow_regime(chains, channels, vb.opt) ? 1u : 0u;
    }
    A.packetBytes = ea.packetBytes;
    A.flags = vb.flags;
    A.flagsF = vb.flags;   // one set unless the launcher overlaps positions
    A.rowReady = nullptr;
    A.ovRowReady = vb.rowReady;
    A.ovFlagsF = vb.flagsF;
    A.ho = vb.ho;
    A.cls = (ClassInfo *)vb.cls;
    A.colChain = vb.colChain;
    A.colsPad = vb.colsPad;
    {
        // ALAC_HIP_SPLIT_CODER=0: one coder wave per 64 chains also in the tiny-batch regime
        const bool split = vb.opt.splitCoder != 0;
        A.bitWordsB = split ? vb.bitWordsB : nullptr;
        A.bitsB = vb.bitsB;
        // the second wave first walks [0, splitAt) keeping only the coder's state (~half the instructions of coding), then
        // codes the rest: both waves finish together at ~2/3 of the frame; whole 48-residual iterations of the coder loop
        A.splitAt = (ea.frameSize * 2 / 3) / 48 * 48;
    }
    A.pubMask = 0;  // producers publish after every tile, rows written through (a release fence per publish cost ~11 us)
    A.flags2 = vb.flags;
    A.dbg = vb.opt.debugWaves ? vb.rowReady : nullptr;  // (the row-ready words are only used by chained tiny batches)
    A.foldDecide = 0;
    // rows that live in the workspace and hold no caller state are never read before they are written: the kernels of the first
    // packet position take init_coefs as constants (load_row) instead of a k_init_state launch writing them first
    A.virgin = (!vb.stateInitialised && vb.stateInternal) ? 1u : 0u;
    if (vb.opt.fastMode && channels == 2) A.virgin = 0;  // no search launch takes init_coefs as constants there: write the rows
    if (!vb.stateInitialised && !A.virgin)
        hipLaunchKernelGGL(k_init_state, dim3((ea.numSegments * 64 + 255) / 256), dim3(256), 0, st, vb.state,
                           ea.numSegments);
#define V1_CASE(D)                                                                                   \
    case D:                                                                                          \
        if (channels == 2)                                                                           \
            launch_v1_typed<D, 2>(A, numPackets, maxSegPackets, st, ev, pa, vs, vb.opt);                 \
        else                                                                                         \
            launch_v1_typed<D, 1>(A, numPackets, maxSegPackets, st, ev, pa, vs, vb.opt);                 \
        break;
    switch (depth) {
        V1_CASE(16)
        V1_CASE(20)
        V1_CASE(24)
        V1_CASE(32)
    default: return hipErrorInvalidValue;
    }
#undef V1_CASE
    return hipGetLastError();
}

}  // namespace alacdev
