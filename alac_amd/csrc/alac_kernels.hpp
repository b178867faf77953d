// alac_kernels.hpp — argument blocks and launchers shared by the kernels and the C-ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alacdev {

// per-channel part of a packet record
struct ChanRec {
    uint32_t bits;      // entropy-coded bits of this channel
    uint16_t num;       // numU / numV
    uint16_t pad;
    int16_t coefs[8];   // header coefficients (row state before the final pass)
};
// one per packet, written by the encode kernel, read by the packer
struct PacketRec {
    uint32_t numSamples;
    uint32_t escape;
    uint32_t mixRes;
    uint32_t totalBits;
    ChanRec c[2];
};
static_assert(sizeof(PacketRec) == 64, "PacketRec layout");

struct EncodeArgs {
    const uint8_t *pcm;
    const uint32_t *numSamples;  // nullable
    const uint32_t *segFirst;    // nullable
    uint32_t numSegments;
    uint32_t frameSize;
    int16_t *state;              // nullable
    int32_t stateIn;
    int32_t *pred;               // [frameSize/8][predStride] search residual scratch
    uint64_t predStride;
    uint32_t *bitWords;          // [packet][2][wcap] per-channel bit strings
    uint32_t wcap;
    PacketRec *recs;
    uint32_t *packetBytes;
    // alac_hip_encode_segmented hands the segment table over UNVALIDATED (no host read-back): every kernel that forms a
    // packet index from it tests the entry against these two bounds (a segment that fails them has no packets), and
    // k_check_segments raises *segBad (device word) for k_finalize / k_scan_sizes / k_pack, which then produce nothing.
    uint32_t numPackets;
    uint32_t segMax;             // longest segment the caller promised (0xffffffff: table validated on the host)
    uint32_t *segBad;            // nullable
};

struct PackArgs {
    const uint8_t *pcm;
    const PacketRec *recs;
    const uint32_t *bitWords;
    uint32_t wcap;
    uint32_t frameSize;
    const uint64_t *offsets;
    uint8_t *out;
    const uint32_t *segBad;      // nullable; != 0: the segment table was refused on the device — pack nothing
};

// Stage timing: when `ev` is non-null, ev[i] is recorded BEFORE stage i and ev[kNumStages] after the
// last one (stages that do not run in a given configuration record back-to-back events).
enum EncodeStage {
    kStageLms1 = 0,   // LPC+mix, mixRes search passes
    kStageGol1,       // Golomb counts of those passes
    kStageLms2,       // LPC+mix, numUV converge passes
    kStageGol2,       // Golomb counts
    kStageLms3,       // LPC+mix, final pass
    kStageGol3,       // Golomb coder, final
    kStageScan,       // finalize + exclusive scan of packet sizes
    kStagePack,       // packer
    kNumStages
};

// lane-per-chain encoder (alac_encode.hip): everything fused in one kernel (recorded as kStageLms3)
hipError_t launch_encode(uint32_t depth, uint32_t channels, const EncodeArgs &ea, const PackArgs &pa,
                         uint32_t numPackets, hipStream_t st, hipEvent_t *ev);
// finalize is the caller's; this launches the size scan and the packer (records ev[kStageScan..])
void launch_scan_pack(uint32_t depth, uint32_t channels, uint32_t *packetBytes, const PackArgs &pa,
                      uint32_t numPackets, hipStream_t st, hipEvent_t *ev, bool recordScan = true);

// tap-parallel pipeline (alac_encode_v1.hip)
// In-launch producer -> consumer hand-offs (alac_encode_v1.hip RowWait, alac_decode_v1.hip k_dec_fused): a consumer
// whose bounded spin runs out has NOT seen its rows.  It raises *err (a host-mapped word the context checks at the next
// synchronize: the call then fails with kALAC_MemFullError instead of returning corrupt packets) and, in the decoder,
// marks its packets kALAC_ParamError.  `lose` is the test switch ALAC_HIP_DEBUG_LOSE_HANDOFF=1: producers never publish.
struct HandoffCtl {
    uint32_t *err = nullptr;
    uint32_t spinLimit = 1u << 22;
    uint32_t lose = 0;
};

// Per-context switches that select code paths (alac_hip_set_option; the ALAC_HIP_* environment variables of the same
// names are only the DEFAULTS a context starts from, read once in alac_hip_create).  -1 = automatic (chosen per call
// from the batch shape).
struct AlacOptions {
    int32_t thru = -1;         // "thru"         ALAC_HIP_THRU        encode: throughput (1) / latency (0) regime, -1 = by batch size
    int32_t narrow = -1;       // "narrow"       ALAC_HIP_NARROW      four lanes per chain, -1 = by batch size (v1_narrow_regime)
    int32_t splitCoder = 1;    // "split_coder"  ALAC_HIP_SPLIT_CODER tiny batches: final coder of a chain on two waves
    int32_t overlapPos = 1;    // "overlap_pos"  ALAC_HIP_OVERLAP_POS chained batches: position p + 1's search beside p's final pass
    int32_t fused = 1;         // "fused"        ALAC_HIP_FUSED       producer/consumer launches (latency regime); 0 = one kernel per stage
    int32_t fold = 1;          // "fold"         ALAC_HIP_FOLD        latency regime: decision and packet sizes inside the final launch
    int32_t fastMode = 0;      // "fast_mode"    (no env)             ALACEncoder::SetFastMode: stereo elements without the search (EncodeStereoFast)
    int32_t laneEncoder = 0;   // "encoder_lane" ALAC_HIP_ENCODER=lane first-generation lane-per-chain encoder
    int32_t laneDecoder = 0;   // "decoder_lane" ALAC_HIP_DECODER=lane first-generation decoder
    int32_t decFused = -1;     // "dec_fused"    ALAC_HIP_DEC_FUSED   decode: entropy wave + its predictor waves in one launch, -1 = by batch size
    int32_t decPair = 1;       // "dec_pair"     ALAC_HIP_DEC_PAIR    decode, separate launches, 16-bit stereo: the predictor lanes of a packet un-mix and write the PCM
    int32_t decDirect = 1;     // "dec_direct"   ALAC_HIP_DEC_DIRECT  decode, separate launches: read the caller's stream directly (no staged
                               // copy): 0 never, 1 from kDecDirectPackets on, 2 whenever legal
    int32_t stageTaps = 1;     // "stage_taps"   ALAC_HIP_STAGE_TAPS  stage-level pc_block: tap-parallel kernel for 5..30 taps
    int32_t loseHandoff = 0;   // "debug_lose_handoff" ALAC_HIP_DEBUG_LOSE_HANDOFF  TEST switch: producers never publish (results invalid by design)
    int32_t debugWaves = 0;    // "debug_waves"  wave placement / timing stamps of the fused final launch into the workspace (tools/wave_map.py)
};
AlacOptions alac_options_from_env();
struct AlacOptionKey {
    const char *name;
    int32_t AlacOptions::*slot;
    int32_t lo, hi;  // accepted values
};
const AlacOptionKey *alac_option_keys(uint32_t *count);
const AlacOptionKey *alac_option_find(const char *key);  // nullptr for an unknown key

struct V1Buffers {
    AlacOptions opt;
    HandoffCtl ho;
    int16_t *state;        // [segments][64] working coefficient rows (caller's d_state or workspace)
    bool stateInitialised; // rows already hold the caller's initial state
    bool stateInternal;    // the rows live in the workspace: nobody reads entries the pipeline does not write
    int32_t *resA, *resB, *resC;
    uint32_t *bits1, *cost2;
    uint32_t *flags;       // progress words of the fused kernels, one per predictor wave (up to chainsPad / 8 + 16)
    uint32_t *flagsF;      // second set (final launch) for the overlapped packet positions of a chained batch
    uint32_t *rowReady;    // [chainsPad] chained batches: packet position + 1 whose final pass has left the chain's 8-tap row
    uint32_t chainsPad;
    void *cls;             // ClassInfo of the class-based final pass (alac_encode_v1.hip)
    uint32_t *colChain;    // [colsPad]
    uint32_t colsPad;      // chainsPad + 256: the two class regions of the final pass are padded to whole waves
    uint32_t *bitWordsB;   // tiny batches: bit words of the second coder wave (same layout as EncodeArgs::bitWords), else null
    uint32_t *bitsB;       // [2 * numPackets + 2]
};
// side stream and fork/join events (owned by the context): the second packet class of the throughput regime's final pass runs
// beside the first, and consecutive packet positions of a chained tiny batch alternate between the two streams
constexpr uint32_t kSideEvents = 2;
struct V1Streams {
    hipStream_t side[1];
    hipEvent_t fork, stagger[kSideEvents], join[kSideEvents];
};
// stage events of one timed call: block 0 = predictor / Golomb stages, block 1 = finalize + scan + pack
constexpr uint32_t kEventBlocks = 2;
// *err = 1 (system scope) unless segFirst[0 .. numSegments] ascends inside [0, numPackets] with no step above maxSeg
void launch_check_segments(const uint32_t *segFirst, uint32_t numSegments, uint32_t numPackets, uint32_t maxSeg, uint32_t *err,
                           uint32_t *segBad,
                           hipStream_t st);
bool v1_throughput_regime(uint32_t numSegments, uint32_t channels, const AlacOptions &opt);
bool v1_narrow_regime(uint64_t chains, uint32_t channels, const AlacOptions &opt);  // four lanes per chain ("tiny") rather than two ("latency")
// ev (nullable): kEventBlocks blocks of kNumStages + 1 events
hipError_t launch_encode_v1(uint32_t depth, uint32_t channels, const EncodeArgs &ea, const PackArgs &pa,
                            const V1Buffers &vb, const V1Streams &vs, uint32_t numPackets, uint32_t maxSegPackets,
                            hipStream_t st, hipEvent_t *ev);

// ---- decode ----
struct DecChan {
    uint16_t mode, denShift, pbFactor, num;
    int16_t coefs[32];
};
struct DecRec {
    uint32_t numSamples;
    uint32_t escape;
    int32_t mixBits, mixRes;
    uint32_t bytesShifted;
    uint32_t elementChannels;  // 1 (SCE/LFE) or 2 (CPE)
    uint64_t shiftPos;         // bit position of the shift-off section inside the packet
    int32_t status;            // of the whole packet in element record 0
    uint32_t pad;              // second-generation decoder: first payload bit of the element
    uint32_t chanIndex;        // first output channel of this element (lane decoder, element sequences)
    uint32_t pad2;
    DecChan c[2];
};

struct DecodeArgs {
    const uint8_t *stream;
    const uint64_t *offsets;
    uint32_t numPackets;
    uint32_t frameSize, bitDepth, numChannels;
    uint32_t mb, pb, kb;
    uint32_t maxElems;  // element records per packet (lane decoder): numChannels, a stream may be all SCE / LFE elements
    DecRec *recs;       // [maxElems][numPackets]
    const uint32_t *gate = nullptr;  // lane decoder as a fallback: its kernels do nothing unless *gate != 0
    HandoffCtl ho;
    int32_t optFused = -1, optPair = 1, optDirect = 1;  // AlacOptions::decFused / decPair / decDirect (host-side launch choices)
    int32_t *resid;  // [ch][frameSize][numPackets] residuals, then samples, in place
    uint8_t *pcmOut;
    uint32_t *numSamplesOut;
    int32_t *statusOut;
};

hipError_t launch_decode(const DecodeArgs &da, hipStream_t st);
// second generation (alac_decode_v1.hip): `words` = capWords uint32 of scratch for the re-staged stream, `plane` =
// numPackets * numChannels * frameSize int32, `prog` = 2 * numPackets + 2 uint32 (progress words of the fused launch;
// chain list and its two counters where the stages are separate launches)
// mismatch (nullable): device counter of the packets whose elements are not the expected sequence (status -4), cleared and
// counted by the pipeline itself; sideStream / fork / join (nullable): the plane and progress-word clears run there beside the
// staging and header kernels
hipError_t launch_decode_v1(const DecodeArgs &da, uint32_t *words, uint64_t capWords, int32_t *plane, uint32_t *prog,
                            hipStream_t st, uint32_t *mismatch = nullptr, hipStream_t sideStream = nullptr, hipEvent_t fork = nullptr,
                            hipEvent_t join = nullptr);

// ---- > 2 channels (alac_multichannel.hip): a packet is a sequence of mono / stereo elements ----
struct McElement {
    uint32_t first;     // channel index of the element's first channel
    uint32_t channels;  // 1 (ID_SCE) or 2 (ID_CPE)
    uint32_t tag;       // element type (3 bits) << 4 | instance tag (4 bits)
};
constexpr uint32_t kMaxChannels = 8;
// the element sequence of a channel count (sChannelMaps, codec/ALACEncoder.cu:97-107); returns the count
uint32_t channel_elements(uint32_t numChannels, McElement *out);
struct McSpliceArgs {
    uint32_t numElements, numPackets;
    McElement el[kMaxChannels];
    const uint8_t *src[kMaxChannels];          // one-element packets of every element, back to back
    const uint64_t *srcOffsets[kMaxChannels];  // [numPackets + 1]
    uint32_t *elemBits;                        // [numElements][numPackets] scratch
    uint32_t *packetBytes;
    uint64_t *offsets;
    uint8_t *out;
};
// channels [first, first + channels) of an interleaved stream -> a compact mono / stereo stream (valid frames only)
void launch_mc_gather(const uint8_t *pcm, uint8_t *out, const uint32_t *numSamples, uint32_t numPackets,
                      uint32_t frameSize, uint32_t numChannels, uint32_t first, uint32_t channels, uint32_t bytesPerSample,
                      hipStream_t st);
// the per-packet sample counts and the segment table of `count` elements batched behind each other
// (sub-packet k * numPackets + p); either input may be null (then its output is not written)
void launch_mc_tables(const uint32_t *numSamples, uint32_t numPackets, const uint32_t *segFirst, uint32_t numSegments,
                      uint32_t count, uint32_t *numSamplesOut, uint32_t *segFirstOut, hipStream_t st);
// sizes + exclusive scan + bit-granular concatenation of the element packets
void launch_mc_splice(const McSpliceArgs &a, hipStream_t st);
void launch_scan_sizes(const uint32_t *sizes, uint64_t *offsets, uint32_t n, hipStream_t st, const uint32_t *segBad = nullptr);

// *count (device) = number of packets whose status is `code`
hipError_t launch_count_status(const int32_t *status, uint32_t n, int32_t code, uint32_t *count, hipStream_t st);

// a stream of 3..8 channels on the second-generation decoder: one pass per element of the channel count's element
// sequence, element r of every packet decoded as the mono / stereo packet that starts where element r - 1 ended
// (elemBit: numPackets uint32 of scratch).  Packets whose elements are not that sequence end with status -4 and are
// counted in *mismatch (device): the caller then decodes the batch with launch_decode, which follows any sequence.
hipError_t launch_decode_v1_elements(const DecodeArgs &da, const McElement *el, uint32_t numElements, uint32_t *words,
                                     uint64_t capWords, int32_t *plane, uint32_t *prog, uint32_t *elemBit,
                                     uint32_t *mismatch, hipStream_t st);

// ---- stage-level ----
hipError_t launch_pc_block(const int32_t *in, int32_t *pc, uint32_t rows, uint32_t stride, int32_t num,
                           int16_t *coefs, int32_t numactive, uint32_t chanbits, uint32_t denshift,
                           bool decode, hipStream_t st, bool allowTaps = true);
// tap-parallel pc_block for any tap count (alac_stage_taps.hip); *_ok tells whether the shape is in its exact range
bool pc_block_taps_ok(int32_t num, int32_t na, uint32_t chanbits, uint32_t denshift);
void launch_pc_block_taps(const int32_t *in, int32_t *pc, uint32_t rows, uint32_t stride, int32_t num, int16_t *coefs,
                          int32_t na, uint32_t chanbits, uint32_t denshift, hipStream_t st);
hipError_t launch_dyn_comp(uint32_t mb0, uint32_t pb, uint32_t kb, const int32_t *pc, uint32_t rows,
                           uint32_t stride, int32_t numSamples, int32_t bitSize, uint8_t *bits,
                           uint32_t bytesStride, uint32_t *numBits, hipStream_t st);
hipError_t launch_dyn_decomp(uint32_t mb0, uint32_t pb, uint32_t kb, const uint8_t *bits,
                             uint32_t bytesStride, uint32_t rows, int32_t *pc, uint32_t stride,
                             int32_t numSamples, int32_t maxSize, uint32_t *numBits, int32_t *status,
                             hipStream_t st);

}  // namespace alacdev
