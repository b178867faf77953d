// alac_capi.hip — the extern "C" boundary of libalac_hip.so (declared in include/alac_hip.h).
// Host-side only: argument checks, workspace carving, kernel launches.  No exceptions cross the
// boundary; every HIP failure becomes one of the reference's int32 status codes.
#include "alac_hip.h"
#include "alac_dev.hpp"
#include "alac_kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace alacdev;

struct alac_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    std::string err;
    // optional per-kernel timing: 4 events per encode call (before encode, after encode, after
    // scan, after pack), recorded on `stream`
    bool profile = false;
    std::vector<hipEvent_t> events;
    uint32_t profCalls = 0;
    std::vector<uint32_t> profSub;  // sub-batches used by each timed call (0 = fused lane encoder)
    // side streams / events of the sub-batch overlap (created on first use)
    V1Streams vs{};
    bool vsReady = false;
    // second stream of the > 2-channel encoder (mono elements beside the stereo ones)
    hipStream_t mcStream = nullptr;
    hipEvent_t mcFork = nullptr, mcJoin = nullptr;
    bool mcReady = false;
    // error word of the in-launch hand-offs (HandoffCtl): pinned host memory the kernels write with a system-scope
    // store when a consumer's bounded wait runs out; read without a copy after a synchronize
    uint32_t *errHost = nullptr;
    uint32_t *errDev = nullptr;
    // code-path switches of this context (alac_hip_set_option); defaults from the ALAC_HIP_* environment at creation
    AlacOptions opt;
};

namespace alacdev {

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

AlacOptions alac_options_from_env()
{
    AlacOptions o;
    o.thru = env_int("ALAC_HIP_THRU", o.thru);
    o.narrow = env_int("ALAC_HIP_NARROW", o.narrow);
    o.splitCoder = env_int("ALAC_HIP_SPLIT_CODER", o.splitCoder) != 0;
    o.overlapPos = env_int("ALAC_HIP_OVERLAP_POS", o.overlapPos) != 0;
    o.fused = env_int("ALAC_HIP_FUSED", o.fused) != 0;
    o.fold = env_int("ALAC_HIP_FOLD", o.fold) != 0;
    if (const char *e = getenv("ALAC_HIP_ENCODER")) o.laneEncoder = strcmp(e, "lane") == 0;
    if (const char *e = getenv("ALAC_HIP_DECODER")) o.laneDecoder = strcmp(e, "lane") == 0;
    if (const char *e = getenv("ALAC_HIP_DEC_FUSED")) o.decFused = *e ? (e[0] == '0' ? 0 : 1) : -1;
    o.decPair = env_int("ALAC_HIP_DEC_PAIR", o.decPair) != 0;
    o.decDirect = env_int("ALAC_HIP_DEC_DIRECT", o.decDirect) != 0;
    o.stageTaps = env_int("ALAC_HIP_STAGE_TAPS", o.stageTaps) != 0;
    o.loseHandoff = env_int("ALAC_HIP_DEBUG_LOSE_HANDOFF", o.loseHandoff) == 1;
    return o;
}

// key -> slot and the range alac_hip_set_option accepts (include/alac_hip.h documents exactly these)
const AlacOptionKey *alac_option_keys(uint32_t *count)
{
    static const AlacOptionKey table[] = {
        {"thru", &AlacOptions::thru, -1, 1},           {"narrow", &AlacOptions::narrow, -1, 1},
        {"split_coder", &AlacOptions::splitCoder, 0, 1}, {"overlap_pos", &AlacOptions::overlapPos, 0, 1},
        {"fused", &AlacOptions::fused, 0, 1},          {"fold", &AlacOptions::fold, 0, 1},
        {"fast_mode", &AlacOptions::fastMode, 0, 1},   {"encoder_lane", &AlacOptions::laneEncoder, 0, 1},
        {"decoder_lane", &AlacOptions::laneDecoder, 0, 1}, {"dec_fused", &AlacOptions::decFused, -1, 1},
        {"dec_pair", &AlacOptions::decPair, 0, 1},     {"dec_direct", &AlacOptions::decDirect, 0, 2},
        {"stage_taps", &AlacOptions::stageTaps, 0, 1},
        {"debug_lose_handoff", &AlacOptions::loseHandoff, 0, 1}, {"debug_waves", &AlacOptions::debugWaves, 0, 1},
    };
    if (count) *count = (uint32_t)(sizeof(table) / sizeof(table[0]));
    return table;
}

const AlacOptionKey *alac_option_find(const char *key)
{
    if (!key) return nullptr;
    uint32_t n = 0;
    const AlacOptionKey *t = alac_option_keys(&n);
    for (uint32_t i = 0; i < n; i++)
        if (strcmp(t[i].name, key) == 0) return t + i;
    return nullptr;
}

}  // namespace alacdev

namespace {

int32_t fail(alac_hip_ctx *ctx, int32_t code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) {
            ctx->err += ": ";
            ctx->err += hipGetErrorString(e);
        }
    }
    return code;
}

bool format_ok(const alac_hip_format *f)
{
    if (!f) return false;
    if (!(f->bit_depth == 16 || f->bit_depth == 20 || f->bit_depth == 24 || f->bit_depth == 32)) return false;
    if (f->num_channels < 1 || f->num_channels > kMaxChannels) return false;
    if (f->frame_size == 0 || f->frame_size > (1u << 20)) return false;
    return true;
}

inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

HandoffCtl handoff_ctl(const alac_hip_ctx *ctx)
{
    HandoffCtl h;
    h.err = ctx->errDev;
    h.lose = ctx->opt.loseHandoff ? 1u : 0u;  // test switch ("debug_lose_handoff"): producers never publish, consumers give up fast
    h.spinLimit = h.lose ? (1u << 8) : (1u << 22);
    return h;
}

// the context's second stream + fork / join events (mono elements beside stereo ones in the > 2-channel encoder; buffer clears
// beside the staging kernels in the decoder), created on first use
bool ensure_second_stream(alac_hip_ctx *ctx)
{
    if (ctx->mcReady) return true;
    if (hipStreamCreateWithFlags(&ctx->mcStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->mcFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->mcJoin, hipEventDisableTiming) != hipSuccess)
        return false;
    ctx->mcReady = true;
    return true;
}

// after the stream has been synchronised: did a consumer of an in-launch hand-off give up?
int32_t check_handoff(alac_hip_ctx *ctx)
{
    if (!ctx->errHost) return ALAC_HIP_noErr;
    volatile uint32_t *w = (volatile uint32_t *)ctx->errHost;
    if (w[1] != 0) {  // k_check_segments: the caller's segment table contradicts the bound it gave
        w[1] = 0;
        w[0] = 0;
        return fail(ctx, ALAC_HIP_ParamError,
                    "d_seg_first is not ascending inside [0, num_packets] or a segment is longer than max_segment_packets: the "
                    "results of the calls since the last synchronize are invalid");
    }
    if (w[0] == 0) return ALAC_HIP_noErr;
    w[0] = 0;
    return fail(ctx, ALAC_HIP_MemFullError,
                "an in-launch producer/consumer hand-off timed out: the results of the calls since the last synchronize are invalid");
}

struct EncLayout {
    uint64_t recs, bitWords, pred, total;
    uint32_t wcap;
    uint64_t predStride;
    // tap-parallel pipeline: residual planes [sample][stream], decision scratch, working state
    uint64_t resA, resB, resC, bits1, cost2, state, flags, rowReady, cls, colChain;
    uint64_t bitWordsB, bitsB;  // tiny batches only (0: absent): second coder wave of the split final coder
    uint64_t segBad;            // one word: k_check_segments refused the caller's segment table (EncodeArgs::segBad)
    uint32_t chainsPad, colsPad;
};

EncLayout enc_layout(const alac_hip_format *f, uint32_t numPackets, uint32_t numSegments)
{
    EncLayout L;
    // escape size bounds what a compressed element may use (codec/ALACEncoder.cu:459,:538)
    const uint64_t escapeBits = (uint64_t)f->frame_size * f->bit_depth * f->num_channels + 32 + 16;
    // + 44 words: the written coder guards its slot 34 words before the end (alac_golomb.hpp, golf_open), and a channel that
    // gets there must by itself be longer than the escape size; a multiple of 4 words: 16-byte aligned channel slots.
    // The slot stride also decides how the coder's scattered stores (one word per lane and symbol in the latency regime,
    // ~10^11 L2 write requests per second) fall onto the L2 channels: measured at 10 000 packets, the final launch takes
    // 0.746 ms with 16 576-byte slots (16-bit stereo, this formula), 0.78-0.80 ms 32 or 256 bytes further, and 0.96 ms with
    // the 24 768 bytes the formula gives 24-bit stereo against 0.76 ms 16 bytes further — hence the extra group there.
    // ALAC_HIP_WCAP_PAD (words) adds to it for experiments.
    L.wcap = (uint32_t)(((escapeBits + 31) / 32 + 44 + 3) & ~3ull);
    if (f->bit_depth == 24) L.wcap += 4;
    if (const char *e = getenv("ALAC_HIP_WCAP_PAD")) L.wcap += (uint32_t)atoi(e) & ~3u;
    const uint64_t lanes = align_up((uint64_t)numSegments * f->num_channels, 64);
    L.predStride = lanes;
    uint64_t off = 0;
    L.recs = off;
    off = align_up(off + (uint64_t)numPackets * sizeof(PacketRec), 256);
    L.bitWords = off;
    // + one spare packet slot: lanes that own no packet write their (ignored) bit words there
    off = align_up(off + ((uint64_t)numPackets + 1) * 2 * L.wcap * 4, 256);
    L.pred = off;
    off = align_up(off + (uint64_t)(f->frame_size / 8 + 1) * lanes * 4, 256);
    L.chainsPad = (uint32_t)lanes;
    // +16 rows: the coder waves read whole 16-row blocks up to the wave's longest chain
    const uint64_t n8 = f->frame_size / 8 + 1 + 16;
    L.resA = off;
    off = align_up(off + (f->num_channels == 2 ? n8 * 5 * lanes * 4 : 0), 256);
    L.resB = off;
    off = align_up(off + n8 * 2 * lanes * 4, 256);
    // final residuals: columns handed out per packet class (k_class_assign), two regions padded to 64 -> up to 128 spare
    L.colsPad = (uint32_t)lanes + 256;  // the two class regions of the final pass are padded to whole waves
    L.resC = off;
    off = align_up(off + ((uint64_t)f->frame_size + 16) * L.colsPad * 4, 256);
    L.bits1 = off;
    off = align_up(off + 5 * lanes * 4, 256);
    L.cost2 = off;
    off = align_up(off + 2 * lanes * 4, 256);
    L.state = off;
    off = align_up(off + (uint64_t)numSegments * 128, 256);
    L.flags = off;  // one progress word per predictor wave (up to lanes / 16), twice: search and final launch of
                    // consecutive packet positions of a chained batch run side by side
    off = align_up(off + 2 * (lanes / 8 + 16) * 4, 256);
    L.rowReady = off;
    off = align_up(off + lanes * 4, 256);
    L.cls = off;  // ClassInfo + per-1024-packet class counts of the compaction
    off = align_up(off + 256 + ((uint64_t)numSegments / 1024 + 2) * 8, 256);
    L.colChain = off;
    off = align_up(off + (uint64_t)L.colsPad * 4, 256);
    // The smallest batches (<= 4096 chains: a chained file, a few hundred files side by side — the low end of the
    // four-lanes-per-chain regime, v1_narrow_regime): the final coder of a chain is split over two waves, the second one
    // writes here
    L.bitWordsB = L.bitsB = 0;
    if (lanes <= 4096) {
        L.bitWordsB = off;
        off = align_up(off + ((uint64_t)numPackets + 1) * 2 * L.wcap * 4, 256);
        L.bitsB = off;
        off = align_up(off + ((uint64_t)numPackets + 1) * 2 * 4, 256);
    }
    L.segBad = off;
    off += 256;
    L.total = off;
    return L;
}

// option "encoder_lane" selects the fused lane-per-chain kernel (alac_encode.hip); default is the
// tap-parallel pipeline (alac_encode_v1.hip).  Both are HIP paths; there is no CPU path.
bool use_lane_encoder(const alac_hip_ctx *ctx) { return ctx->opt.laneEncoder != 0; }

// > 2 channels: the mono / stereo pipeline once per element over a gathered copy of its channels, then the splice
// (alac_multichannel.hip)
alac_hip_format element_format(const alac_hip_format *f, uint32_t channels)
{
    alac_hip_format e = *f;
    e.num_channels = channels;
    return e;
}

uint64_t max_output_bytes(const alac_hip_format *fmt, uint32_t num_packets)
{
    // escape elements: 7 + 16 + 32 + N*ch*depth bits each, + ID_END, rounded up; +8 so word stores may overhang
    const uint64_t elems = fmt->num_channels > 2 ? fmt->num_channels : 1;
    const uint64_t per = ((uint64_t)fmt->frame_size * fmt->num_channels * fmt->bit_depth + 55 * elems + 3 + 7) / 8;
    return align_up(per * num_packets + 8, 16);
}

// the elements of one type form one batch of count * numPackets one-element packets (sub-packet k * P + p)
struct McGroup {
    uint32_t channels, count, elem[kMaxChannels];
    uint64_t gather, sub, subBytes, out, outCap, sizes, offs, ns, seg, state;
};

struct McLayout {
    uint32_t numElements;
    McElement el[kMaxChannels];
    uint32_t groupOf[kMaxChannels], indexInGroup[kMaxChannels];
    McGroup g[2];  // [0] stereo elements, [1] mono elements
    uint64_t elemBits, total;
};

McLayout mc_layout(const alac_hip_format *f, uint32_t numPackets, uint32_t numSegments)
{
    McLayout M;
    M.numElements = channel_elements(f->num_channels, M.el);
    M.g[0].channels = 2;
    M.g[1].channels = 1;
    M.g[0].count = M.g[1].count = 0;
    for (uint32_t e = 0; e < M.numElements; e++) {
        McGroup &G = M.g[M.el[e].channels == 2 ? 0 : 1];
        M.groupOf[e] = M.el[e].channels == 2 ? 0 : 1;
        M.indexInGroup[e] = G.count;
        G.elem[G.count++] = e;
    }
    uint64_t off = 0;
    for (int gi = 0; gi < 2; gi++) {
        McGroup &G = M.g[gi];
        const alac_hip_format gf = element_format(f, G.channels);
        const uint64_t subPackets = (uint64_t)G.count * numPackets, subSegments = (uint64_t)G.count * numSegments;
        G.gather = off;
        off = align_up(off + subPackets * f->frame_size * G.channels * bytes_per_sample(f->bit_depth) + 64, 256);
        G.sub = off;
        G.subBytes = G.count ? enc_layout(&gf, (uint32_t)subPackets, (uint32_t)subSegments).total : 0;
        off = align_up(off + G.subBytes, 256);
        G.outCap = max_output_bytes(&gf, (uint32_t)subPackets);
        G.out = off;
        off = align_up(off + G.outCap + 16, 256);
        G.sizes = off;
        off = align_up(off + subPackets * 4, 256);
        G.offs = off;
        off = align_up(off + (subPackets + 1) * 8, 256);
        G.ns = off;
        off = align_up(off + subPackets * 4, 256);
        G.seg = off;
        off = align_up(off + (subSegments + 1) * 4, 256);
        G.state = off;
        off = align_up(off + subSegments * ALAC_HIP_STATE_INT16 * 2, 256);
    }
    M.elemBits = off;
    off = align_up(off + (uint64_t)M.numElements * numPackets * 4, 256);
    M.total = off;
    return M;
}

struct DecLayout {
    uint64_t recs, resid, words, capWords, prog, elemBit, mismatch, total;
    uint32_t maxElems;
};

// streamBytes = 0: every packet at its largest regular size (what an encoder of this library or Apple's can emit);
// a stream padded with ID_FIL / ID_DSE elements may be longer, and a caller who knows its length passes it
DecLayout dec_layout(const alac_hip_format *f, uint32_t numPackets, uint64_t streamBytes = 0)
{
    DecLayout L;
    uint64_t off = 0;
    L.recs = off;
    // a mono / stereo stream may still be a sequence of SCE / LFE elements (codec/ALACDecoder.cu:622-756): the lane
    // decoder keeps one record per channel
    L.maxElems = f->num_channels;
    off = align_up(off + (uint64_t)numPackets * L.maxElems * sizeof(DecRec), 256);
    L.resid = off;
    off = align_up(off + (uint64_t)f->num_channels * f->frame_size * numPackets * 4 + 256, 256);  // + block over-read
    L.prog = off;  // progress words of the fused launch / chain list, counters and pair list of the separate launches
    off = align_up(off + (uint64_t)numPackets * 28 + 64, 256);  // dec_lists (alac_decode_v1.hip): 7 n words + 16 counters
    L.elemBit = off;
    off = align_up(off + (uint64_t)numPackets * 4, 256);
    L.mismatch = off;
    off = align_up(off + 4, 256);
    // LAST: the stream re-staged as MSB-first words + 64 zero words (alac_decode_v1.hip).  Whatever the caller's
    // workspace holds beyond this offset is used, so a longer stream only needs a larger workspace.
    L.words = off;
    const uint64_t regular = (uint64_t)numPackets * max_output_bytes(f, 1);
    L.capWords = ((streamBytes > regular ? streamBytes : regular) + 3) / 4 + 64;
    off = align_up(off + L.capWords * 4, 256);
    L.total = off;
    return L;
}

bool use_lane_decoder(const alac_hip_ctx *ctx) { return ctx->opt.laneDecoder != 0; }

struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(uint64_t n) { return hipMalloc(&p, n ? n : 4); }
};

}  // namespace

extern "C" {

int32_t alac_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int32_t alac_hip_create(alac_hip_ctx **out_ctx, int32_t device, void *stream)
{
    if (!out_ctx) return ALAC_HIP_ParamError;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return ALAC_HIP_ParamError;
    alac_hip_ctx *c = new (std::nothrow) alac_hip_ctx;
    if (!c) return ALAC_HIP_MemFullError;
    c->device = device;
    c->opt = alac_options_from_env();
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return ALAC_HIP_ParamError;
    }
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return ALAC_HIP_MemFullError;
        }
        c->ownStream = true;
    }
    if (hipHostMalloc((void **)&c->errHost, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->errDev, c->errHost, 0) != hipSuccess) {
        alac_hip_destroy(c);
        return ALAC_HIP_MemFullError;
    }
    c->errHost[0] = c->errHost[1] = 0;
    *out_ctx = c;
    return ALAC_HIP_noErr;
}

void alac_hip_destroy(alac_hip_ctx *ctx)
{
    if (!ctx) return;
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    ctx->events.clear();
    if (ctx->vsReady) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->vs.side[0]);
        (void)hipStreamDestroy(ctx->vs.side[0]);
        for (uint32_t i = 0; i < kSideEvents; i++) {
            (void)hipEventDestroy(ctx->vs.stagger[i]);
            (void)hipEventDestroy(ctx->vs.join[i]);
        }
        (void)hipEventDestroy(ctx->vs.fork);
    }
    if (ctx->mcReady) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->mcStream);
        (void)hipStreamDestroy(ctx->mcStream);
        (void)hipEventDestroy(ctx->mcFork);
        (void)hipEventDestroy(ctx->mcJoin);
    }
    if (ctx->ownStream && ctx->stream) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->errHost) (void)hipHostFree(ctx->errHost);
    delete ctx;
}

int32_t alac_hip_synchronize(alac_hip_ctx *ctx)
{
    if (!ctx) return ALAC_HIP_ParamError;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipStreamSynchronize", e);
    return check_handoff(ctx);
}

int32_t alac_hip_set_option(alac_hip_ctx *ctx, const char *key, int32_t value)
{
    if (!ctx) return ALAC_HIP_ParamError;
    const AlacOptionKey *k = alac_option_find(key);
    if (!k) return fail(ctx, ALAC_HIP_ParamError, "unknown option");
    if (value < k->lo || value > k->hi) return fail(ctx, ALAC_HIP_ParamError, "option value outside its documented range");
    ctx->opt.*(k->slot) = value;
    return ALAC_HIP_noErr;
}

int32_t alac_hip_get_option(alac_hip_ctx *ctx, const char *key, int32_t *value)
{
    if (!ctx || !value) return ALAC_HIP_ParamError;
    const AlacOptionKey *k = alac_option_find(key);
    if (!k) return fail(ctx, ALAC_HIP_ParamError, "unknown option");
    *value = ctx->opt.*(k->slot);
    return ALAC_HIP_noErr;
}

uint64_t alac_hip_debug_waves_offset(const alac_hip_format *fmt, uint32_t num_packets, uint32_t num_segments)
{
    if (!format_ok(fmt) || fmt->num_channels > 2) return 0;
    return enc_layout(fmt, num_packets, num_segments ? num_segments : num_packets).rowReady;
}

const char *alac_hip_encode_regime(alac_hip_ctx *ctx, const alac_hip_format *fmt, uint32_t num_segments)
{
    if (!ctx || !format_ok(fmt)) return "";
    if (use_lane_encoder(ctx)) return "lane";
    const uint32_t ch = fmt->num_channels > 2 ? 2 : fmt->num_channels;
    if (v1_throughput_regime(num_segments, ch, ctx->opt)) return "throughput";
    const uint64_t chains = (uint64_t)num_segments * ch;
    // the launcher's own predicates (launch_encode_v1 / launch_v1_typed): fuse = fused && !thru, narrow = narrow && fuse
    if (!ctx->opt.fused) return "stagewise";
    const bool narrow = v1_narrow_regime(chains, ch, ctx->opt);
    return narrow ? "tiny" : "latency";
}

const char *alac_hip_last_error(const alac_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void *alac_hip_stream(const alac_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

uint64_t alac_hip_encode_workspace_bytes(const alac_hip_format *fmt, uint32_t num_packets, uint32_t num_segments)
{
    if (!format_ok(fmt)) return 0;
    if (fmt->num_channels > 2) return mc_layout(fmt, num_packets, num_segments ? num_segments : num_packets).total;
    return enc_layout(fmt, num_packets, num_segments ? num_segments : num_packets).total;
}

uint32_t alac_hip_state_int16(const alac_hip_format *fmt)
{
    if (!format_ok(fmt)) return 0;
    McElement el[kMaxChannels];
    return ALAC_HIP_STATE_INT16 * (fmt->num_channels > 2 ? channel_elements(fmt->num_channels, el) : 1);
}

uint64_t alac_hip_encode_max_output_bytes(const alac_hip_format *fmt, uint32_t num_packets)
{
    if (!format_ok(fmt)) return 0;
    return max_output_bytes(fmt, num_packets);
}

static int32_t encode_elements(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                               const uint32_t *d_num_samples, uint32_t num_packets, const uint32_t *d_seg_first,
                               uint32_t num_segments, int16_t *d_state, int32_t state_in, void *d_workspace,
                               uint64_t workspace_bytes, uint8_t *d_out, uint64_t out_capacity,
                               uint32_t *d_packet_bytes, uint64_t *d_packet_offsets, uint32_t maxSegHint);

// one mono / stereo batch; `timed` = this call may consume a slot of the armed stage timing
static int32_t encode_core(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                           const uint32_t *d_num_samples, uint32_t num_packets, const uint32_t *d_seg_first,
                           uint32_t num_segments, int16_t *d_state, int32_t state_in, void *d_workspace,
                           uint64_t workspace_bytes, uint8_t *d_out, uint64_t out_capacity,
                           uint32_t *d_packet_bytes, uint64_t *d_packet_offsets, bool timed, uint32_t maxSegHint);

int32_t alac_hip_encode_segmented(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                                  const uint32_t *d_num_samples, uint32_t num_packets, const uint32_t *d_seg_first,
                                  uint32_t num_segments, uint32_t max_segment_packets, int16_t *d_state, int32_t state_in,
                                  void *d_workspace, uint64_t workspace_bytes, uint8_t *d_out, uint64_t out_capacity,
                                  uint32_t *d_packet_bytes, uint64_t *d_packet_offsets)
{
    if (!ctx) return ALAC_HIP_ParamError;
    if (!format_ok(fmt)) return fail(ctx, ALAC_HIP_ParamError, "unsupported format");
    if (fmt->num_channels > 2)
        return encode_elements(ctx, fmt, d_pcm, d_num_samples, num_packets, d_seg_first, num_segments, d_state, state_in,
                               d_workspace, workspace_bytes, d_out, out_capacity, d_packet_bytes, d_packet_offsets,
                               max_segment_packets);
    return encode_core(ctx, fmt, d_pcm, d_num_samples, num_packets, d_seg_first, num_segments, d_state, state_in,
                       d_workspace, workspace_bytes, d_out, out_capacity, d_packet_bytes, d_packet_offsets, true,
                       max_segment_packets);
}

int32_t alac_hip_encode(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                        const uint32_t *d_num_samples, uint32_t num_packets, const uint32_t *d_seg_first,
                        uint32_t num_segments, int16_t *d_state, int32_t state_in, void *d_workspace,
                        uint64_t workspace_bytes, uint8_t *d_out, uint64_t out_capacity,
                        uint32_t *d_packet_bytes, uint64_t *d_packet_offsets)
{
    return alac_hip_encode_segmented(ctx, fmt, d_pcm, d_num_samples, num_packets, d_seg_first, num_segments, 0, d_state, state_in,
                                     d_workspace, workspace_bytes, d_out, out_capacity, d_packet_bytes, d_packet_offsets);
}

static int32_t encode_elements(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                               const uint32_t *d_num_samples, uint32_t num_packets, const uint32_t *d_seg_first,
                               uint32_t num_segments, int16_t *d_state, int32_t state_in, void *d_workspace,
                               uint64_t workspace_bytes, uint8_t *d_out, uint64_t out_capacity,
                               uint32_t *d_packet_bytes, uint64_t *d_packet_offsets, uint32_t maxSegHint)
{
    if (num_packets == 0) return ALAC_HIP_noErr;
    if (!d_pcm || !d_workspace || !d_out || !d_packet_bytes || !d_packet_offsets)
        return fail(ctx, ALAC_HIP_ParamError, "null buffer");
    if (!d_seg_first) num_segments = num_packets;
    if (num_segments == 0 || num_segments > num_packets) return fail(ctx, ALAC_HIP_ParamError, "bad segment count");
    if (((uintptr_t)d_workspace & 255)) return fail(ctx, ALAC_HIP_ParamError, "misaligned workspace (256 B)");
    if ((uint64_t)num_packets * kMaxChannels > 0x7fffffffull) return fail(ctx, ALAC_HIP_ParamError, "too many packets");
    const McLayout M = mc_layout(fmt, num_packets, num_segments);
    if (workspace_bytes < M.total) return fail(ctx, ALAC_HIP_ParamError, "workspace too small");
    if (out_capacity < max_output_bytes(fmt, num_packets))
        return fail(ctx, ALAC_HIP_ParamError, "output capacity below alac_hip_encode_max_output_bytes");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipSetDevice");
    uint8_t *ws = (uint8_t *)d_workspace;
    const uint32_t bps = bytes_per_sample(fmt->bit_depth);
    hipStream_t mainStream = ctx->stream;

    // the stereo batch runs on the context's stream, the mono batch beside it on a second stream (both are bound by
    // the latency of one wave, not by the machine)
    const bool both = M.g[0].count && M.g[1].count;
    bool side = both;
    if (side && !ensure_second_stream(ctx)) return fail(ctx, ALAC_HIP_MemFullError, "creating the second stream");
    if (side) {
        (void)hipEventRecord(ctx->mcFork, mainStream);
        (void)hipStreamWaitEvent(ctx->mcStream, ctx->mcFork, 0);
    }
    int32_t rc = ALAC_HIP_noErr;
    hipError_t copyErr = hipSuccess;
    // SetFastMode is consulted for 2-channel STREAMS only (codec/ALACEncoder.cu:998-1001): the stereo elements of a
    // > 2-channel stream are searched like everything else
    const int32_t fastModeOfCtx = ctx->opt.fastMode;
    ctx->opt.fastMode = 0;
    struct Restore {
        alac_hip_ctx *c;
        int32_t v;
        ~Restore() { c->opt.fastMode = v; }
    } restoreFast{ctx, fastModeOfCtx};
    for (int gi = 0; gi < 2 && rc == ALAC_HIP_noErr; gi++) {
        const McGroup &G = M.g[gi];
        if (!G.count) continue;
        hipStream_t st = (side && gi == 1) ? ctx->mcStream : mainStream;
        const alac_hip_format gf = element_format(fmt, G.channels);
        const uint64_t elemPcm = (uint64_t)num_packets * fmt->frame_size * G.channels * bps;
        for (uint32_t k = 0; k < G.count; k++)
            launch_mc_gather((const uint8_t *)d_pcm, ws + G.gather + k * elemPcm, d_num_samples, num_packets, fmt->frame_size,
                             fmt->num_channels, M.el[G.elem[k]].first, G.channels, bps, st);
        uint32_t *ns = d_num_samples ? (uint32_t *)(ws + G.ns) : nullptr;
        uint32_t *seg = d_seg_first ? (uint32_t *)(ws + G.seg) : nullptr;
        launch_mc_tables(d_num_samples, num_packets, d_seg_first, num_segments, G.count, ns, seg, st);
        // coefficient rows: the caller's [element][segment][64] <-> the batch's [k][segment][64]
        int16_t *gstate = d_state ? (int16_t *)(ws + G.state) : nullptr;
        const uint64_t rowBytes = (uint64_t)num_segments * ALAC_HIP_STATE_INT16 * 2;
        if (gstate && state_in)
            for (uint32_t k = 0; k < G.count && copyErr == hipSuccess; k++)
                copyErr = hipMemcpyAsync((uint8_t *)gstate + k * rowBytes, (const uint8_t *)d_state + G.elem[k] * rowBytes,
                                         rowBytes, hipMemcpyDeviceToDevice, st);
        ctx->stream = st;
        rc = encode_core(ctx, &gf, ws + G.gather, ns, G.count * num_packets, seg, G.count * num_segments, gstate, state_in,
                         ws + G.sub, G.subBytes, ws + G.out, G.outCap, (uint32_t *)(ws + G.sizes), (uint64_t *)(ws + G.offs),
                         false, maxSegHint);
        ctx->stream = mainStream;
        if (gstate && rc == ALAC_HIP_noErr)
            for (uint32_t k = 0; k < G.count && copyErr == hipSuccess; k++)
                copyErr = hipMemcpyAsync((uint8_t *)d_state + G.elem[k] * rowBytes, (const uint8_t *)gstate + k * rowBytes,
                                         rowBytes, hipMemcpyDeviceToDevice, st);
    }
    if (side) {
        (void)hipEventRecord(ctx->mcJoin, ctx->mcStream);
        (void)hipStreamWaitEvent(mainStream, ctx->mcJoin, 0);
    }
    // (the second stream is joined above whatever happened, so the context's stream stays the only one to wait on)
    if (rc != ALAC_HIP_noErr) return rc;
    if (copyErr != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "coefficient state copy", copyErr);
    McSpliceArgs sa;
    sa.numElements = M.numElements;
    sa.numPackets = num_packets;
    for (uint32_t e = 0; e < M.numElements; e++) {
        const McGroup &G = M.g[M.groupOf[e]];
        sa.el[e] = M.el[e];
        sa.src[e] = ws + G.out;
        sa.srcOffsets[e] = (const uint64_t *)(ws + G.offs) + (uint64_t)M.indexInGroup[e] * num_packets;
    }
    sa.elemBits = (uint32_t *)(ws + M.elemBits);
    sa.packetBytes = d_packet_bytes;
    sa.offsets = d_packet_offsets;
    sa.out = d_out;
    launch_mc_splice(sa, mainStream);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "splice launch", e);
    return ALAC_HIP_noErr;
}

static int32_t encode_core(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *d_pcm,
                           const uint32_t *d_num_samples, uint32_t num_packets, const uint32_t *d_seg_first,
                           uint32_t num_segments, int16_t *d_state, int32_t state_in, void *d_workspace,
                           uint64_t workspace_bytes, uint8_t *d_out, uint64_t out_capacity,
                           uint32_t *d_packet_bytes, uint64_t *d_packet_offsets, bool timed, uint32_t maxSegHint)
{
    if (num_packets == 0) return ALAC_HIP_noErr;
    if (!d_pcm || !d_workspace || !d_out || !d_packet_bytes || !d_packet_offsets)
        return fail(ctx, ALAC_HIP_ParamError, "null buffer");
    if (!d_seg_first) num_segments = num_packets;
    if (num_segments == 0 || num_segments > num_packets) return fail(ctx, ALAC_HIP_ParamError, "bad segment count");
    if (((uintptr_t)d_out & 3) || ((uintptr_t)d_workspace & 255) || ((uintptr_t)d_pcm & 15))
        return fail(ctx, ALAC_HIP_ParamError, "misaligned buffer (out 4 B, pcm 16 B, workspace 256 B)");
    const EncLayout L = enc_layout(fmt, num_packets, num_segments);
    if (workspace_bytes < L.total) return fail(ctx, ALAC_HIP_ParamError, "workspace too small");
    if (out_capacity < alac_hip_encode_max_output_bytes(fmt, num_packets))
        return fail(ctx, ALAC_HIP_ParamError, "output capacity below alac_hip_encode_max_output_bytes");

    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipSetDevice");
    uint8_t *ws = (uint8_t *)d_workspace;
    EncodeArgs ea;
    ea.pcm = (const uint8_t *)d_pcm;
    ea.numSamples = d_num_samples;
    ea.segFirst = d_seg_first;
    ea.numSegments = num_segments;
    ea.frameSize = fmt->frame_size;
    ea.state = d_state;
    ea.stateIn = state_in;
    ea.pred = (int32_t *)(ws + L.pred);
    ea.predStride = L.predStride;
    ea.bitWords = (uint32_t *)(ws + L.bitWords);
    ea.wcap = L.wcap;
    ea.recs = (PacketRec *)(ws + L.recs);
    ea.packetBytes = d_packet_bytes;
    // a table with the caller's bound (alac_hip_encode_segmented) is not read back: the kernels test every entry they use
    const bool unread = d_seg_first && maxSegHint;
    ea.numPackets = num_packets;
    ea.segMax = unread ? (maxSegHint < num_packets ? maxSegHint : num_packets) : 0xffffffffu;
    ea.segBad = unread ? (uint32_t *)(ws + L.segBad) : nullptr;
    PackArgs pa;
    pa.pcm = ea.pcm;
    pa.recs = ea.recs;
    pa.bitWords = ea.bitWords;
    pa.wcap = L.wcap;
    pa.frameSize = fmt->frame_size;
    pa.offsets = d_packet_offsets;
    pa.out = d_out;
    pa.segBad = ea.segBad;
    constexpr uint32_t EV = kEventBlocks * (kNumStages + 1);
    hipEvent_t *ev = nullptr;
    if (timed && ctx->profile && (uint64_t)(ctx->profCalls + 1) * EV <= ctx->events.size())
        ev = &ctx->events[ctx->profCalls++ * EV];
    hipError_t e;
    // the caller's bound on the segment length (alac_hip_encode_segmented) is checked on the device whatever kernels run
    if (unread)
        launch_check_segments(d_seg_first, num_segments, num_packets, ea.segMax, ctx->errDev ? ctx->errDev + 1 : nullptr,
                              ea.segBad, ctx->stream);
    if (use_lane_encoder(ctx)) {
        if (ev) ctx->profSub.push_back(0);
        e = launch_encode(fmt->bit_depth, fmt->num_channels, ea, pa, num_packets, ctx->stream,
                          ev ? ev + (kNumStages + 1) : nullptr);
    } else {
        if (!ctx->vsReady) {
            bool ok = hipEventCreateWithFlags(&ctx->vs.fork, hipEventDisableTiming) == hipSuccess &&
                      hipStreamCreateWithFlags(&ctx->vs.side[0], hipStreamNonBlocking) == hipSuccess;
            for (uint32_t i = 0; ok && i < kSideEvents; i++)
                ok = hipEventCreateWithFlags(&ctx->vs.stagger[i], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&ctx->vs.join[i], hipEventDisableTiming) == hipSuccess;
            if (!ok) return fail(ctx, ALAC_HIP_MemFullError, "creating side streams");
            ctx->vsReady = true;
        }
        if (ev) ctx->profSub.push_back(1);
        // packets per segment: the pipeline runs once per packet position (a chained segment is serial)
        uint32_t maxSeg = 1;
        if (d_seg_first && maxSegHint) {
            // the caller knows how long its longest segment is (alac_hip_encode_segmented): no read-back of the table, no
            // host wait.  The bound is checked on the device; a table that contradicts it fails the next synchronize.
            maxSeg = maxSegHint < num_packets ? maxSegHint : num_packets;
        } else if (d_seg_first) {
            std::vector<uint32_t> sf(num_segments + 1);
            if (hipMemcpyAsync(sf.data(), d_seg_first, (num_segments + 1) * 4ull, hipMemcpyDeviceToHost, ctx->stream) !=
                    hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                return fail(ctx, ALAC_HIP_ParamError, "reading d_seg_first");
            maxSeg = 0;
            for (uint32_t s = 0; s < num_segments; s++) {
                if (sf[s + 1] < sf[s] || sf[s + 1] > num_packets) return fail(ctx, ALAC_HIP_ParamError, "bad d_seg_first");
                maxSeg = sf[s + 1] - sf[s] > maxSeg ? sf[s + 1] - sf[s] : maxSeg;
            }
        }
        V1Buffers vb;
        vb.opt = ctx->opt;
        vb.state = d_state ? d_state : (int16_t *)(ws + L.state);
        vb.stateInitialised = d_state && state_in;
        vb.stateInternal = d_state == nullptr;
        vb.resA = (int32_t *)(ws + L.resA);
        vb.resB = (int32_t *)(ws + L.resB);
        vb.resC = (int32_t *)(ws + L.resC);
        vb.bits1 = (uint32_t *)(ws + L.bits1);
        vb.cost2 = (uint32_t *)(ws + L.cost2);
        vb.flags = (uint32_t *)(ws + L.flags);
        vb.flagsF = vb.flags + (L.chainsPad / 8 + 16);
        vb.rowReady = (uint32_t *)(ws + L.rowReady);
        vb.chainsPad = L.chainsPad;
        vb.cls = ws + L.cls;
        vb.colChain = (uint32_t *)(ws + L.colChain);
        vb.colsPad = L.colsPad;
        vb.bitWordsB = L.bitWordsB ? (uint32_t *)(ws + L.bitWordsB) : nullptr;
        vb.bitsB = L.bitsB ? (uint32_t *)(ws + L.bitsB) : nullptr;
        vb.ho = handoff_ctl(ctx);
        e = launch_encode_v1(fmt->bit_depth, fmt->num_channels, ea, pa, vb, ctx->vs, num_packets, maxSeg, ctx->stream, ev);
    }
    if (e != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "encode launch", e);
    return ALAC_HIP_noErr;
}

uint32_t alac_hip_num_stages(void) { return kNumStages; }

const char *alac_hip_stage_name(uint32_t stage)
{
    static const char *names[kNumStages] = {"lms_search1", "golomb_count1", "lms_search2", "golomb_count2",
                                            "lms_final",   "golomb_final",  "finalize_scan", "pack"};
    return stage < kNumStages ? names[stage] : "";
}

int32_t alac_hip_profile_begin(alac_hip_ctx *ctx, uint32_t max_calls)
{
    if (!ctx) return ALAC_HIP_ParamError;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipSetDevice");
    while (ctx->events.size() < (size_t)max_calls * kEventBlocks * (kNumStages + 1)) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return fail(ctx, ALAC_HIP_MemFullError, "hipEventCreate");
        ctx->events.push_back(e);
    }
    ctx->profCalls = 0;
    ctx->profSub.clear();
    ctx->profile = max_calls != 0;
    return ALAC_HIP_noErr;
}

int32_t alac_hip_profile_end(alac_hip_ctx *ctx, uint32_t *out_calls, float *out_stage_ms, uint32_t *out_launches)
{
    if (!ctx) return ALAC_HIP_ParamError;
    ctx->profile = false;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipStreamSynchronize", e);
    constexpr uint32_t BLK = kNumStages + 1;
    constexpr uint32_t EV = kEventBlocks * BLK;
    double t[kNumStages] = {0};
    double launches[kNumStages] = {0};
    auto elapsed = [&](hipEvent_t a, hipEvent_t b, double &acc) -> bool {
        float ms = 0;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return false;
        acc += ms;
        return true;
    };
    for (uint32_t c = 0; c < ctx->profCalls; c++) {
        hipEvent_t *base = &ctx->events[(size_t)c * EV];
        const uint32_t H = ctx->profSub[c];
        for (uint32_t h = 0; h < H; h++)  // predictor / Golomb stages, one launch per sub-batch
            for (uint32_t k = kStageLms1; k <= kStageGol3; k++) {
                if (!elapsed(base[h * BLK + k], base[h * BLK + k + 1], t[k]))
                    return fail(ctx, ALAC_HIP_ParamError, "hipEventElapsedTime");
                launches[k] += 1;
            }
        hipEvent_t *tail = base + BLK;
        for (uint32_t k = (H ? kStageScan : 0); k < kNumStages; k++) {
            if (!elapsed(tail[k], tail[k + 1], t[k])) return fail(ctx, ALAC_HIP_ParamError, "hipEventElapsedTime");
            launches[k] += 1;
        }
    }
    if (out_calls) *out_calls = ctx->profCalls;
    const double n = ctx->profCalls ? ctx->profCalls : 1;
    for (uint32_t k = 0; k < kNumStages; k++) {
        if (out_stage_ms) out_stage_ms[k] = launches[k] > 0 ? (float)(t[k] / launches[k]) : 0.0f;  // per launch
        if (out_launches) out_launches[k] = (uint32_t)(launches[k] / n + 0.5);                   // per call
    }
    return ALAC_HIP_noErr;
}

uint32_t alac_hip_magic_cookie(const alac_hip_format *fmt, uint32_t max_frame_bytes, uint32_t avg_bit_rate,
                               uint8_t *c)
{
    if (!format_ok(fmt) || !c) return 0;
    auto be32 = [](uint8_t *p, uint32_t v) {
        p[0] = (uint8_t)(v >> 24);
        p[1] = (uint8_t)(v >> 16);
        p[2] = (uint8_t)(v >> 8);
        p[3] = (uint8_t)v;
    };
    be32(c + 0, fmt->frame_size);
    c[4] = 0;  // kALACCompatibleVersion
    c[5] = (uint8_t)fmt->bit_depth;
    c[6] = (uint8_t)kPB0;
    c[7] = (uint8_t)kMB0;
    c[8] = (uint8_t)kKB0;
    c[9] = (uint8_t)fmt->num_channels;
    c[10] = 0;
    c[11] = 255;  // MAX_RUN_DEFAULT
    be32(c + 12, max_frame_bytes);
    be32(c + 16, avg_bit_rate);
    be32(c + 20, fmt->sample_rate);
    return 24;
}

uint32_t alac_hip_magic_cookie_size(const alac_hip_format *fmt)
{
    if (!format_ok(fmt)) return 0;
    return fmt->num_channels > 2 ? 48 : 24;  // + kChannelAtomSize 12 + sizeof(ALACAudioChannelLayout) 12
}

uint32_t alac_hip_magic_cookie_full(const alac_hip_format *fmt, uint32_t max_frame_bytes, uint32_t avg_bit_rate,
                                    uint8_t *c, uint32_t capacity)
{
    const uint32_t need = alac_hip_magic_cookie_size(fmt);
    if (!need || !c || capacity < need) return 0;  // "no incomplete cookies", codec/ALACEncoder.cu:1136-1139
    alac_hip_magic_cookie(fmt, max_frame_bytes, avg_bit_rate, c);
    if (need == 24) return 24;
    // ALACChannelLayoutTags, codec/ALACAudioTypes.h:103-124
    static const uint32_t tags[kMaxChannels] = {(100u << 16) | 1, (101u << 16) | 2, (113u << 16) | 3, (116u << 16) | 4,
                                                (120u << 16) | 5, (124u << 16) | 6, (142u << 16) | 7, (127u << 16) | 8};
    memset(c + 24, 0, 24);
    c[27] = 24;  // theChannelAtom[3] = sizeof(ALACAudioChannelLayout) + kChannelAtomSize
    memcpy(c + 28, "chan", 4);
    // the fork stores mChannelLayoutTag in host byte order (:1120 has no Swap32NtoB): little endian
    const uint32_t t = tags[fmt->num_channels - 1];
    c[36] = (uint8_t)t;
    c[37] = (uint8_t)(t >> 8);
    c[38] = (uint8_t)(t >> 16);
    c[39] = (uint8_t)(t >> 24);
    return 48;
}

int32_t alac_hip_format_from_cookie(const uint8_t *ck, uint32_t size, alac_hip_format *out)
{
    if (!ck || !out) return ALAC_HIP_ParamError;
    // ALACDecoder::Init skips legacy 'frma' and 'alac' atoms (codec/ALACDecoder.cu:123-134)
    if (size >= 12 && ck[4] == 'f' && ck[5] == 'r' && ck[6] == 'm' && ck[7] == 'a') {
        ck += 12;
        size -= 12;
    }
    if (size >= 12 && ck[4] == 'a' && ck[5] == 'l' && ck[6] == 'a' && ck[7] == 'c') {
        ck += 12;
        size -= 12;
    }
    if (size < 24) return ALAC_HIP_ParamError;
    if (ck[4] > 0) return ALAC_HIP_ParamError;  // compatibleVersion <= kALACVersion (:153)
    out->frame_size = ((uint32_t)ck[0] << 24) | ((uint32_t)ck[1] << 16) | ((uint32_t)ck[2] << 8) | ck[3];
    out->bit_depth = ck[5];
    out->num_channels = ck[9];
    out->sample_rate = ((uint32_t)ck[20] << 24) | ((uint32_t)ck[21] << 16) | ((uint32_t)ck[22] << 8) | ck[23];
    return ALAC_HIP_noErr;
}

uint64_t alac_hip_decode_workspace_bytes(const alac_hip_format *fmt, uint32_t num_packets)
{
    if (!format_ok(fmt)) return 0;
    return dec_layout(fmt, num_packets).total;
}

uint64_t alac_hip_decode_workspace_bytes_stream(const alac_hip_format *fmt, uint32_t num_packets, uint64_t stream_bytes)
{
    if (!format_ok(fmt)) return 0;
    return dec_layout(fmt, num_packets, stream_bytes).total;
}

int32_t alac_hip_decode(alac_hip_ctx *ctx, const uint8_t *h_cookie, uint32_t cookie_size, const uint8_t *d_stream,
                        const uint64_t *d_packet_offsets, uint32_t num_packets, void *d_workspace,
                        uint64_t workspace_bytes, uint8_t *d_pcm_out, uint32_t *d_num_samples_out,
                        int32_t *d_status)
{
    if (!ctx) return ALAC_HIP_ParamError;
    alac_hip_format fmt;
    if (alac_hip_format_from_cookie(h_cookie, cookie_size, &fmt) != ALAC_HIP_noErr)
        return fail(ctx, ALAC_HIP_ParamError, "bad magic cookie");
    if (!format_ok(&fmt)) return fail(ctx, ALAC_HIP_ParamError, "unsupported format in cookie");
    if (num_packets == 0) return ALAC_HIP_noErr;
    if (!d_stream || !d_packet_offsets || !d_workspace || !d_pcm_out || !d_num_samples_out || !d_status)
        return fail(ctx, ALAC_HIP_ParamError, "null buffer");
    if (((uintptr_t)d_workspace & 255) || ((uintptr_t)d_pcm_out & 3))
        return fail(ctx, ALAC_HIP_ParamError, "misaligned buffer");
    DecLayout L = dec_layout(&fmt, num_packets);
    if (workspace_bytes < L.total) return fail(ctx, ALAC_HIP_ParamError, "workspace too small");
    L.capWords = (workspace_bytes - L.words) / 4;  // all of it: packets that do not fit the staged words get status -50
    // skip wrappers again to reach pb/mb/kb
    const uint8_t *ck = h_cookie;
    uint32_t size = cookie_size;
    if (size >= 12 && ck[4] == 'f' && ck[5] == 'r' && ck[6] == 'm' && ck[7] == 'a') { ck += 12; size -= 12; }
    if (size >= 12 && ck[4] == 'a' && ck[5] == 'l' && ck[6] == 'a' && ck[7] == 'c') { ck += 12; size -= 12; }
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipSetDevice");
    uint8_t *ws = (uint8_t *)d_workspace;
    DecodeArgs da;
    da.stream = d_stream;
    da.offsets = d_packet_offsets;
    da.numPackets = num_packets;
    da.frameSize = fmt.frame_size;
    da.bitDepth = fmt.bit_depth;
    da.numChannels = fmt.num_channels;
    da.pb = ck[6];
    da.mb = ck[7];
    da.kb = ck[8];
    // Local hardening, NOT reference behaviour (codec/ag_dec.c:282-286 only checks its pointers; ALACDecoder::Init takes any
    // pb / mb / kb): the kernels shift by kb (k = min(lg3a, kb), m = (1 << k) - 1), so kb outside 1..16 would be an undefined
    // or zero-width shift, and pb = 0 freezes the mean at values the quotient code does not expect.  No encoder writes either.
    if (da.kb < 1 || da.kb > 16 || da.pb == 0) return fail(ctx, ALAC_HIP_ParamError, "bad AG parameters in cookie (pb / kb)");
    da.maxElems = L.maxElems;
    da.recs = (DecRec *)(ws + L.recs);
    da.resid = (int32_t *)(ws + L.resid);
    da.pcmOut = d_pcm_out;
    da.numSamplesOut = d_num_samples_out;
    da.statusOut = d_status;
    da.ho = handoff_ctl(ctx);
    da.optFused = ctx->opt.decFused;
    da.optPair = ctx->opt.decPair;
    da.optDirect = ctx->opt.decDirect;
    hipError_t e;
    if (use_lane_decoder(ctx)) {
        e = launch_decode(da, ctx->stream);
    } else if (fmt.num_channels > 2) {
        // one pass per element of the channel count's sequence (the position of element k + 1 is only known once
        // element k is entropy-decoded); a stream with another sequence is decoded again by the lane decoder, which
        // follows whatever the packets carry.  This is the one place where the call waits for the GPU.
        McElement el[kMaxChannels];
        const uint32_t nel = channel_elements(fmt.num_channels, el);
        e = launch_decode_v1_elements(da, el, nel, (uint32_t *)(ws + L.words), L.capWords, da.resid, (uint32_t *)(ws + L.prog),
                                      (uint32_t *)(ws + L.elemBit), (uint32_t *)(ws + L.mismatch), ctx->stream);
        uint32_t mismatch = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&mismatch, ws + L.mismatch, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && mismatch) e = launch_decode(da, ctx->stream);
    } else {
        // (the plane / progress clears on the context's second stream beside the staging and header kernels were measured in
        // round 3: 1.978 against 1.952 ms per 10 000 packets — the two cross-stream waits cost more than the 45 us of clears
        // they hide; launch_decode_v1 keeps the parameters)
        const bool side = false;
        // A mono / stereo stream whose packets carry another element sequence (two SCEs for two channels, fill in
        // front of ...: status -4 from the fast pipeline, counted by its header kernel) is decoded again by the lane decoder,
        // which follows whatever the packets carry.  No host round trip: its kernels are gated on that device-side count.
        e = launch_decode_v1(da, (uint32_t *)(ws + L.words), L.capWords, da.resid, (uint32_t *)(ws + L.prog), ctx->stream,
                             (uint32_t *)(ws + L.mismatch), side ? ctx->mcStream : nullptr, ctx->mcFork, ctx->mcJoin);
        if (e == hipSuccess) {
            DecodeArgs dg = da;
            dg.gate = (const uint32_t *)(ws + L.mismatch);
            e = launch_decode(dg, ctx->stream);
        }
    }
    if (e != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "decode launch", e);
    return ALAC_HIP_noErr;
}

// ---- stage level ---------------------------------------------------------------------------------

int32_t alac_hip_pc_block(alac_hip_ctx *ctx, const int32_t *d_in, int32_t *d_pc, uint32_t num_rows,
                          uint32_t row_stride, int32_t num, int16_t *d_coefs, int32_t numactive, uint32_t chanbits,
                          uint32_t denshift)
{
    if (!ctx || !d_in || !d_pc) return ALAC_HIP_ParamError;
    if (numactive < 0 || numactive > 31 || chanbits < 1 || chanbits > 32 || denshift > 15 || num < 0)
        return fail(ctx, ALAC_HIP_ParamError, "bad pc_block parameters");
    if (numactive != 0 && numactive != 31 && !d_coefs) return fail(ctx, ALAC_HIP_ParamError, "null coefs");
    hipError_t e = launch_pc_block(d_in, d_pc, num_rows, row_stride, num, d_coefs, numactive, chanbits, denshift,
                                   false, ctx->stream, ctx->opt.stageTaps != 0);
    return e == hipSuccess ? ALAC_HIP_noErr : fail(ctx, ALAC_HIP_ParamError, "pc_block launch", e);
}

int32_t alac_hip_unpc_block(alac_hip_ctx *ctx, const int32_t *d_pc, int32_t *d_out, uint32_t num_rows,
                            uint32_t row_stride, int32_t num, int16_t *d_coefs, int32_t numactive,
                            uint32_t chanbits, uint32_t denshift)
{
    if (!ctx || !d_pc || !d_out) return ALAC_HIP_ParamError;
    if (numactive < 0 || numactive > 31 || chanbits < 1 || chanbits > 32 || denshift > 15 || num < 0)
        return fail(ctx, ALAC_HIP_ParamError, "bad unpc_block parameters");
    if (numactive != 0 && numactive != 31 && !d_coefs) return fail(ctx, ALAC_HIP_ParamError, "null coefs");
    hipError_t e = launch_pc_block(d_pc, d_out, num_rows, row_stride, num, d_coefs, numactive, chanbits, denshift,
                                   true, ctx->stream);
    return e == hipSuccess ? ALAC_HIP_noErr : fail(ctx, ALAC_HIP_ParamError, "unpc_block launch", e);
}

int32_t alac_hip_dyn_comp(alac_hip_ctx *ctx, uint32_t mb0, uint32_t pb, uint32_t kb, const int32_t *d_pc,
                          uint32_t num_rows, uint32_t row_stride, int32_t num_samples, int32_t bit_size,
                          uint8_t *d_bits, uint32_t bytes_stride, uint32_t *d_num_bits)
{
    if (!ctx || !d_pc || !d_num_bits) return ALAC_HIP_ParamError;
    if (bit_size < 1 || bit_size > 32 || kb < 1 || kb > 16 || num_samples < 0)  // codec/ag_enc.c:268
        return fail(ctx, ALAC_HIP_ParamError, "bad dyn_comp parameters");
    if (d_bits && ((bytes_stride & 3) || ((uintptr_t)d_bits & 3)))
        return fail(ctx, ALAC_HIP_ParamError, "bit buffer must be 4-byte aligned/strided");
    hipError_t e = launch_dyn_comp(mb0, pb, kb, d_pc, num_rows, row_stride, num_samples, bit_size, d_bits,
                                   bytes_stride, d_num_bits, ctx->stream);
    return e == hipSuccess ? ALAC_HIP_noErr : fail(ctx, ALAC_HIP_ParamError, "dyn_comp launch", e);
}

int32_t alac_hip_dyn_decomp(alac_hip_ctx *ctx, uint32_t mb0, uint32_t pb, uint32_t kb, const uint8_t *d_bits,
                            uint32_t bytes_stride, uint32_t num_rows, int32_t *d_pc, uint32_t row_stride,
                            int32_t num_samples, int32_t max_size, uint32_t *d_num_bits, int32_t *d_status)
{
    if (!ctx || !d_bits || !d_pc || !d_num_bits || !d_status) return ALAC_HIP_ParamError;
    if (max_size < 1 || max_size > 32 || kb < 1 || kb > 16 || num_samples < 0)
        return fail(ctx, ALAC_HIP_ParamError, "bad dyn_decomp parameters");
    hipError_t e = launch_dyn_decomp(mb0, pb, kb, d_bits, bytes_stride, num_rows, d_pc, row_stride, num_samples,
                                     max_size, d_num_bits, d_status, ctx->stream);
    return e == hipSuccess ? ALAC_HIP_noErr : fail(ctx, ALAC_HIP_ParamError, "dyn_decomp launch", e);
}

// ---- host-buffer convenience ---------------------------------------------------------------------

int32_t alac_hip_encode_host_segments(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *h_pcm,
                                      const uint32_t *h_num_samples, uint32_t num_packets, const uint32_t *h_seg_first,
                                      uint32_t num_segments, int16_t *h_state, int32_t state_in, uint8_t *h_out,
                                      uint64_t out_capacity, uint32_t *h_packet_bytes, uint64_t *out_total_bytes)
{
    if (!ctx) return ALAC_HIP_ParamError;
    if (!format_ok(fmt)) return fail(ctx, ALAC_HIP_ParamError, "unsupported format");
    if (out_total_bytes) *out_total_bytes = 0;
    if (num_packets == 0) return ALAC_HIP_noErr;
    if (!h_pcm || !h_out || !h_packet_bytes || !h_num_samples || !h_seg_first || num_segments == 0)
        return fail(ctx, ALAC_HIP_ParamError, "null buffer");
    if (h_seg_first[0] != 0 || h_seg_first[num_segments] != num_packets)
        return fail(ctx, ALAC_HIP_ParamError, "segment table must start at 0 and end at num_packets");
    for (uint32_t s = 0; s < num_segments; s++)
        if (h_seg_first[s] > h_seg_first[s + 1]) return fail(ctx, ALAC_HIP_ParamError, "segment table not ascending");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipSetDevice");

    const uint32_t bpf = fmt->num_channels * bytes_per_sample(fmt->bit_depth);
    const uint32_t np = num_packets, nseg = num_segments;
    const uint64_t stateBytes = (uint64_t)nseg * alac_hip_state_int16(fmt) * 2;
    const uint64_t pcmBytes = (uint64_t)np * fmt->frame_size * bpf;
    const uint64_t wsBytes = alac_hip_encode_workspace_bytes(fmt, np, nseg);
    const uint64_t outMax = alac_hip_encode_max_output_bytes(fmt, np);

    DevBuf dPcm, dNs, dSeg, dState, dWs, dOut, dSizes, dOffs;
    hipError_t e;
    if ((e = dPcm.alloc(pcmBytes)) || (e = dNs.alloc(np * 4ull)) || (e = dSeg.alloc((nseg + 1) * 4ull)) ||
        (e = dState.alloc(stateBytes)) || (e = dWs.alloc(wsBytes)) || (e = dOut.alloc(outMax)) ||
        (e = dSizes.alloc(np * 4ull)) || (e = dOffs.alloc((np + 1) * 8ull)))
        return fail(ctx, ALAC_HIP_MemFullError, "hipMalloc", e);
    hipStream_t st = ctx->stream;
    if ((e = hipMemcpyAsync(dPcm.p, h_pcm, pcmBytes, hipMemcpyHostToDevice, st)) ||
        (e = hipMemcpyAsync(dNs.p, h_num_samples, np * 4ull, hipMemcpyHostToDevice, st)) ||
        (e = hipMemcpyAsync(dSeg.p, h_seg_first, (nseg + 1) * 4ull, hipMemcpyHostToDevice, st)))
        return fail(ctx, ALAC_HIP_ParamError, "H2D copy", e);
    if (h_state && state_in)
        if ((e = hipMemcpyAsync(dState.p, h_state, stateBytes, hipMemcpyHostToDevice, st)))
            return fail(ctx, ALAC_HIP_ParamError, "H2D state", e);
    uint32_t maxSeg = 1;
    for (uint32_t s2 = 0; s2 < nseg; s2++) maxSeg = h_seg_first[s2 + 1] - h_seg_first[s2] > maxSeg ? h_seg_first[s2 + 1] - h_seg_first[s2] : maxSeg;
    int32_t rc = alac_hip_encode_segmented(ctx, fmt, dPcm.p, (const uint32_t *)dNs.p, np, (const uint32_t *)dSeg.p, nseg, maxSeg,
                                           (int16_t *)dState.p, (h_state && state_in) ? 1 : 0, dWs.p, wsBytes,
                                           (uint8_t *)dOut.p, outMax, (uint32_t *)dSizes.p, (uint64_t *)dOffs.p);
    if (rc != ALAC_HIP_noErr) return rc;
    uint64_t total = 0;
    if ((e = hipMemcpyAsync(&total, (uint64_t *)dOffs.p + np, 8, hipMemcpyDeviceToHost, st)) ||
        (e = hipStreamSynchronize(st)))
        return fail(ctx, ALAC_HIP_ParamError, "encode execution", e);
    if (total > out_capacity) return fail(ctx, ALAC_HIP_MemFullError, "host output buffer too small");
    if ((e = hipMemcpyAsync(h_out, dOut.p, total, hipMemcpyDeviceToHost, st)) ||
        (e = hipMemcpyAsync(h_packet_bytes, dSizes.p, np * 4ull, hipMemcpyDeviceToHost, st)))
        return fail(ctx, ALAC_HIP_ParamError, "D2H copy", e);
    if (h_state)
        if ((e = hipMemcpyAsync(h_state, dState.p, stateBytes, hipMemcpyDeviceToHost, st)))
            return fail(ctx, ALAC_HIP_ParamError, "D2H state", e);
    if ((e = hipStreamSynchronize(st))) return fail(ctx, ALAC_HIP_ParamError, "sync", e);
    if (int32_t hrc = check_handoff(ctx)) return hrc;
    if (out_total_bytes) *out_total_bytes = total;
    return ALAC_HIP_noErr;
}

int32_t alac_hip_encode_host(alac_hip_ctx *ctx, const alac_hip_format *fmt, const void *h_pcm,
                             uint64_t total_samples, uint32_t segment_packets, int16_t *h_state, int32_t state_in,
                             uint8_t *h_out, uint64_t out_capacity, uint32_t *h_packet_bytes,
                             uint64_t *out_total_bytes)
{
    if (!ctx) return ALAC_HIP_ParamError;
    if (!format_ok(fmt)) return fail(ctx, ALAC_HIP_ParamError, "unsupported format");
    if (out_total_bytes) *out_total_bytes = 0;
    if (total_samples == 0) return ALAC_HIP_noErr;
    if (!h_pcm || !h_out || !h_packet_bytes) return fail(ctx, ALAC_HIP_ParamError, "null buffer");
    const uint32_t bpf = fmt->num_channels * bytes_per_sample(fmt->bit_depth);
    const uint64_t np64 = (total_samples + fmt->frame_size - 1) / fmt->frame_size;
    if (np64 > 0x7fffffffull) return fail(ctx, ALAC_HIP_ParamError, "too many packets");
    const uint32_t np = (uint32_t)np64;
    const uint32_t nseg = segment_packets ? (np + segment_packets - 1) / segment_packets : 1;
    std::vector<uint32_t> ns(np, fmt->frame_size), segFirst(nseg + 1);
    ns[np - 1] = (uint32_t)(total_samples - (uint64_t)(np - 1) * fmt->frame_size);
    for (uint32_t s = 0; s <= nseg; s++) {
        uint64_t f = segment_packets ? (uint64_t)s * segment_packets : (s ? np : 0);
        segFirst[s] = (uint32_t)(f < np ? f : np);
    }
    // the last packet may be partial: hand over whole packets (zero padded)
    const uint64_t inBytes = total_samples * bpf, pcmBytes = (uint64_t)np * fmt->frame_size * bpf;
    const void *src = h_pcm;
    std::vector<uint8_t> padded;
    if (inBytes != pcmBytes) {
        padded.assign(pcmBytes, 0);
        memcpy(padded.data(), h_pcm, inBytes);
        src = padded.data();
    }
    return alac_hip_encode_host_segments(ctx, fmt, src, ns.data(), np, segFirst.data(), nseg, h_state, state_in, h_out,
                                         out_capacity, h_packet_bytes, out_total_bytes);
}

int32_t alac_hip_decode_host(alac_hip_ctx *ctx, const uint8_t *h_cookie, uint32_t cookie_size,
                             const uint8_t *h_stream, const uint32_t *h_packet_bytes, uint32_t num_packets,
                             uint8_t *h_pcm_out, uint32_t *h_num_samples_out, int32_t *h_status)
{
    if (!ctx) return ALAC_HIP_ParamError;
    alac_hip_format fmt;
    if (alac_hip_format_from_cookie(h_cookie, cookie_size, &fmt) != ALAC_HIP_noErr || !format_ok(&fmt))
        return fail(ctx, ALAC_HIP_ParamError, "bad magic cookie");
    if (num_packets == 0) return ALAC_HIP_noErr;
    if (!h_stream || !h_packet_bytes || !h_pcm_out || !h_num_samples_out || !h_status)
        return fail(ctx, ALAC_HIP_ParamError, "null buffer");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, ALAC_HIP_ParamError, "hipSetDevice");
    std::vector<uint64_t> offs(num_packets + 1, 0);
    for (uint32_t i = 0; i < num_packets; i++) offs[i + 1] = offs[i] + h_packet_bytes[i];
    const uint64_t total = offs[num_packets];
    const uint32_t bpf = fmt.num_channels * bytes_per_sample(fmt.bit_depth);
    const uint64_t pcmBytes = (uint64_t)num_packets * fmt.frame_size * bpf;
    const uint64_t wsBytes = alac_hip_decode_workspace_bytes_stream(&fmt, num_packets, total);
    DevBuf dStream, dOffs, dWs, dPcm, dNs, dSt;
    hipError_t e;
    if ((e = dStream.alloc(total + 16)) || (e = dOffs.alloc((num_packets + 1) * 8ull)) || (e = dWs.alloc(wsBytes)) ||
        (e = dPcm.alloc(pcmBytes)) || (e = dNs.alloc(num_packets * 4ull)) || (e = dSt.alloc(num_packets * 4ull)))
        return fail(ctx, ALAC_HIP_MemFullError, "hipMalloc", e);
    hipStream_t st = ctx->stream;
    if ((e = hipMemcpyAsync(dStream.p, h_stream, total, hipMemcpyHostToDevice, st)) ||
        (e = hipMemcpyAsync(dOffs.p, offs.data(), (num_packets + 1) * 8ull, hipMemcpyHostToDevice, st)) ||
        (e = hipMemsetAsync(dPcm.p, 0, pcmBytes, st)))
        return fail(ctx, ALAC_HIP_ParamError, "H2D copy", e);
    int32_t rc = alac_hip_decode(ctx, h_cookie, cookie_size, (const uint8_t *)dStream.p, (const uint64_t *)dOffs.p,
                                 num_packets, dWs.p, wsBytes, (uint8_t *)dPcm.p, (uint32_t *)dNs.p,
                                 (int32_t *)dSt.p);
    if (rc != ALAC_HIP_noErr) return rc;
    if ((e = hipMemcpyAsync(h_pcm_out, dPcm.p, pcmBytes, hipMemcpyDeviceToHost, st)) ||
        (e = hipMemcpyAsync(h_num_samples_out, dNs.p, num_packets * 4ull, hipMemcpyDeviceToHost, st)) ||
        (e = hipMemcpyAsync(h_status, dSt.p, num_packets * 4ull, hipMemcpyDeviceToHost, st)) ||
        (e = hipStreamSynchronize(st)))
        return fail(ctx, ALAC_HIP_ParamError, "decode execution", e);
    return check_handoff(ctx);
}

}  // extern "C"
