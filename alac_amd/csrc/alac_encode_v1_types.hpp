// alac_encode_v1_types.hpp — argument block of the tap-parallel encode pipeline (alac_encode_v1_impl.hpp) and the
// per-depth launcher the dispatcher in alac_encode_v1.hip calls; the launcher is instantiated once per bit depth in its own
// translation unit (alac_encode_v1_d16/20/24/32.hip) so that the build runs them side by side.
#pragma once
#include "alac_dev.hpp"
#include "alac_kernels.hpp"

namespace alacdev {

struct SegView {
    const uint8_t *pcm;
    const uint32_t *numSamples;
    const uint32_t *segFirst;
    uint32_t numSegments, frameSize, pos;
    uint32_t segBegin, segEnd;  // the sub-batch [segBegin, segEnd) this launch works on
    uint32_t numPackets, segMax;  // bounds every entry of an unvalidated segFirst is tested against (EncodeArgs::segMax)
};

constexpr uint32_t kNoChain = 0xffffffffu;

// What k_class_count / k_class_assign leave for the final pass: the packets that still need it (not escaped, present at this packet
// position), grouped by the widest predictor they chose.  Columns [0, n8) carry the chains of packets with an 8-tap
// channel, [base4, base4 + n4) the chains of all-4-tap packets; both regions are padded to whole coder waves (64).
// A packet's channels sit in adjacent columns, so the staging still mixes U and V from one PCM load.
struct ClassInfo {
    uint32_t n8, n4;     // chains (columns in use) per class
    uint32_t base4;      // first column of the 4-tap region = n8 rounded up to 64
    uint32_t nCols;      // base4 + n4 rounded up to 64
};

struct V1Args {
    SegView S;
    int16_t *state;        // [segment][64] working coefficient rows
    PacketRec *recs;
    int32_t *resA;         // search1 residuals [j < n8][5 * chainsPad]   stream = r * chainsPad + chain
    int32_t *resB;         // search2 residuals [j < n8][2 * chainsPad]   stream = rowsel * chainsPad + chain
    int32_t *resC;         // final residuals   [j < N][chainsPad]
    uint32_t *bits1;       // [5 * chainsPad]
    uint32_t *cost2;       // [2 * chainsPad]
    uint32_t chainsPad;    // numSegments * channels rounded up to 64
    uint32_t *bitWords;
    uint32_t wcap;
    uint32_t dumpSlot;     // channel slot index (2 * numPackets) no packet owns: bit words of lanes without a packet
    uint32_t *packetBytes;
    uint32_t *flags;       // producer progress words of the fused search launch (zeroed per call)
    uint32_t *flags2;      // ... of the fused converge launch (k_search2_fused): behind those of the search launch
    uint32_t *dbg;         // option "debug_waves": 8 dwords per workgroup of the fused final launch (null: off)
    uint32_t virgin;       // 1: the coefficient rows have never been written (first packet position, state in the workspace):
                           // load_row takes init_coefs instead of reading them
    uint32_t foldDecide;   // 1: k_final_fused<.., FOLD> decides numU / numV / escape and the packet size itself (no k_decide2,
                           // no k_finalize launch)
    uint32_t pubMask;      // producers publish after every (low byte + 1) tiles; bit 31: with a release fence
    uint32_t idleFast;     // 1: lanes without work do not force the checked paths (latency regime, see launcher)
    HandoffCtl ho;         // error word / spin bound / test switch of the in-launch hand-offs
    uint32_t thru;         // 1: throughput regime (see launch_v1_typed)
    uint32_t narrow;       // 1: tiny batch: four lanes per chain
    // chained tiny batches: the mixRes search of packet position p + 1 runs beside the final pass of position p
    uint32_t *rowReady;    // [chains] position + 1 whose final pass has stored the chain's 8-tap row (0: not used)
    uint32_t *flagsF;      // progress words of the final launch (the search launch next to it uses `flags`)
    uint32_t *ovRowReady, *ovFlagsF;  // the buffers the launcher switches the two above to when it overlaps positions
    // final pass by packet class (k_class_count, k_class_assign): columns of the residual plane are handed out per class
    ClassInfo *cls;
    uint32_t *colChain;    // [colsPad] chain (segment * CH + channel) of every column, kNoChain for pad columns
    uint32_t colsPad;      // row stride of resC in the class layout
    // coder split over two waves (tiny batches, k_final_fused): the second wave's bit words and bit counts
    uint32_t *bitWordsB;   // same layout as bitWords; null: no split
    uint32_t *bitsB;       // [2 * numPackets + 2] bits the second wave wrote per channel slot
    uint32_t splitAt;      // residuals [0, splitAt) belong to the first wave
};

// search + final passes of every packet position, then finalize / scan / pack (alac_encode_v1_impl.hpp); explicitly
// instantiated for DEPTH in {16, 20, 24, 32} x CH in {1, 2}
template <int DEPTH, int CH>
void launch_v1_typed(const V1Args &A0, uint32_t numPackets, uint32_t maxSegPackets, hipStream_t st, hipEvent_t *ev,
                     const PackArgs &pa, const V1Streams &vs, const AlacOptions &opt);

}  // namespace alacdev
