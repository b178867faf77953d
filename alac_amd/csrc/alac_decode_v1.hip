// alac_decode_v1.hip — batch ALAC decode, second generation (the default; ALAC_HIP_DECODER=lane selects the
// first one in alac_decode.hip).
//
//   k_dec_stage      copy of the packed stream into 32-bit words, MSB first (byte swapped), zero padded: the
//                    bit readers below then work on aligned words and may over-read safely
//   k_dec_header     one lane per packet: element / header parse (codec/ALACDecoder.cu:571-1002): record, payload
//                    position, status; who writes the packet's PCM (DecRec::pad2); in the separate-launch regime the
//                    work lists of the kernels below (dec_lists)
//   entropy lanes    dyn_decomp (codec/ag_dec.c:272-362) of U and V, one lane per packet, from a 64-bit register bit
//                    window refilled out of a per-lane LDS ring that is re-staged from the word stream every 16
//                    residuals with 16-byte loads issued a round ahead.  A packet is serial by nature (V starts where
//                    U's bits end): the parallelism is the batch, and a residual costs 55 instructions.
//   predictor        unpc_block (codec/dp_dec.c:55-381): two lanes per chain behind the entropy lanes (unpc_fast_body),
//                    one lane per chain for large batches (unpc_wide_body, unpc_pair_body), generic lane-serial
//                    kernel for everything else (k_dec_unpc)
//   k_dec_raw        uncompressed elements: fixed-width fields, one thread per sample-frame
//   k_dec_unmix      un-mix + shift-byte re-attach + PCM packing (codec/ALACDecoder.cu:193-563) for the packets whose
//                    PCM nobody else wrote
// Two launch shapes (decode_v1_pass): up to 65 536 chains ONE launch k_dec_fused_wg — a workgroup is the entropy wave of
// 48 packets and the three predictor waves that follow it through its rows (workgroup-scope hand-off, LDS progress
// words); above that k_dec_raw, k_dec_entropy_wide, k_dec_unpc_wide as separate launches that each fill the machine.
//
// Sample planes are CHAIN-major here: int32 [packet][channel][frameSize], so every lane streams through its own
// contiguous row (the entropy lanes drift apart on zero runs, which rules out the sample-major layout of the
// encoder), and the un-mix reads both channels of a packet coalesced along the sample index.
#include <cstdlib>
#include "alac_dev.hpp"
#include "alac_kernels.hpp"
#include "alac_unpc.hpp"
#include "alac_lms.hpp"

namespace alacdev {

constexpr uint32_t kFastChanBits = 23;  // widest sample the __mul24 predictor bodies take (see k_dec_header)
constexpr int kDecRound = 16;    // symbols decoded between two restagings of the LDS ring
constexpr int kWinWords = 16;    // words staged per lane per round (64 bytes)
constexpr int kHdrWinWords = 17;  // k_dec_header: words of a packet's head kept in LDS per lane (64 bytes at any byte phase)
constexpr int kHdrWinStride = 17; // odd: conflict-free columns
constexpr int kWinStride = 33;   // LDS words per lane: a circular ring of 32 (+1: odd stride, conflict-free columns)

// ---- k_dec_stage ---------------------------------------------------------------------------------
// words[i] = bytes 4i .. 4i+3 of the stream as one MSB-first word, followed by 64 words of zeros
// zeroA[0 .. nA) and zeroB[0]: the pass's device counters (work-list counters, mismatch count).  As hipMemsetAsync calls in front
// of the first kernel they were two blit launches with a 30-50 us bubble in front of the first (rocprofv3 kernel trace of
// back-to-back decode passes, round 4); the first kernel of the pass clears them instead.
struct DecZero {
    uint32_t *a, *b;
    uint32_t nA;
};
__device__ __forceinline__ void dec_zero(const DecZero &z)
{
    if (blockIdx.x != 0) return;
    if (z.a && threadIdx.x < z.nA) z.a[threadIdx.x] = 0;
    if (z.b && threadIdx.x == 0) z.b[0] = 0;
}
__global__ __launch_bounds__(256) void k_dec_stage(const uint8_t *stream, const uint64_t *offsets, uint32_t numPackets,
                                                   uint32_t *words, uint64_t capWords, DecZero zero)
{
    dec_zero(zero);
    const uint64_t total = offsets[numPackets];
    const uint64_t fullWords = total >> 2;
    const bool aligned = ((uintptr_t)stream & 3) == 0;
    const uint64_t endWords = min(capWords, ((total + 3) >> 2) + 64);  // the readers never go past this (dec_word_limit)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < endWords; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        if (i < fullWords && aligned) {
            v = __builtin_bswap32(((const uint32_t *)stream)[i]);
        } else if (i * 4 < total) {
            for (uint32_t b = 0; b < 4; b++) {
                const uint64_t bi = i * 4 + b;
                v = (v << 8) | (bi < total ? stream[bi] : 0u);
            }
        }
        words[i] = v;
    }
}

// ---- per-lane bit window ---------------------------------------------------------------------------
// Two consecutive stream words in registers plus the bit offset into the first; the third word is always already
// read from the LDS ring (its latency is covered by the symbols decoded before it is needed).  Skipping is
// branch-free: the words move down under a select when the offset crosses 32.
struct BitWin {
    uint32_t w0, w1;     // words idx-2 and idx-1 of the lane's word stream
    uint32_t o;          // bits of w0 already consumed (0..31)
    uint32_t idx;        // index of the look-ahead word, relative to the word that was staged into ring slot 0
    uint32_t nw;         // look-ahead word, ring[idx & 31]
    const uint32_t *ring;
};

__device__ __forceinline__ uint32_t bw_pos(const BitWin &b) { return (b.idx - 2u) * 32u + b.o; }

__device__ __forceinline__ uint32_t bw_peek(const BitWin &b)
{
    // one 64-bit shift (v_lshlrev_b64) instead of the shift-by-zero-safe funnel of two 32-bit words (four instructions)
    return (uint32_t)(((((uint64_t)b.w0 << 32) | b.w1) << b.o) >> 32);
}

__device__ __forceinline__ void bw_skip(BitWin &b, uint32_t n)  // n <= 32
{
    const uint32_t t = b.o + n;
    const bool carry = t >= 32u;
    b.o = t & 31u;
    b.w0 = carry ? b.w1 : b.w0;
    b.w1 = carry ? b.nw : b.w1;
    b.idx += carry ? 1u : 0u;
    b.nw = b.ring[b.idx & 31u];
}

__device__ __forceinline__ uint32_t bw_get(BitWin &b, uint32_t n)  // 1 <= n <= 32
{
    const uint32_t v = bw_peek(b) >> (32 - n);
    bw_skip(b, n);
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ uint32_t wave_min_dec(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

struct DecV1Args {
    DecodeArgs d;
    const uint32_t *words;   // staged stream (MSB-first words + 64 zero words, k_dec_stage) — or, DIRECT (raw != null), unused
    // DIRECT (round 4, the separate launches of a large batch): the kernels read the CALLER's stream as little-endian dwords
    // and swap them as they load (one v_perm_b32 per word) instead of reading a staged copy — k_dec_stage's pass over the whole
    // stream (1.7 GB of traffic, 0.35 ms at 125 000 packets) is gone.  Word i comes from raw[i] while i < tailStart, from the
    // small copy tail[i - tailStart] behind that (k_dec_tail: the stream's last 16 whole dwords, its ragged end and zeros), so
    // that no load ever touches a byte behind the caller's buffer; tailStart = max(0, (total bytes >> 2) - 16) is computed on
    // the device (the total lives there).
    const uint32_t *raw;     // null: staged
    const uint32_t *tail;    // kDecTailWords words
    uint64_t capWords;
    int32_t *plane;          // [packet][channel][frameSize]
    uint32_t *prog;          // fused launch: [packet][2] residuals completed per channel (0xffffffff = all)
    uint32_t pubMask;        // the entropy lanes publish every (pubMask + 1) rounds of 16 symbols
    // element rounds of a > 2-channel stream (launch_decode_v1_elements): round r decodes element r of every packet
    // as a mono / stereo packet that starts elemBit[p] bits into packet p and leaves the element's end there
    uint32_t *elemBit;       // null for mono / stereo streams
    uint32_t round;
    uint32_t outChannels;    // interleaved channels of the output frame (= element channels for mono / stereo)
    uint32_t outFirst;       // output channel of the element's first channel
    uint32_t *mismatch;      // null, or: k_dec_header counts the packets it gives status -4 (another element sequence) here
    uint32_t lists;          // separate launches: k_dec_header sorts the packets / chains into work lists (dec_lists below)
    uint32_t pairs;          // ... and lists the packets whose two chains unpc_pair_body takes (option dec_pair, one-lane predictor)
};

constexpr uint32_t kDecTailWords = 128;  // 16 whole dwords + a ragged one + the 64 zero words the readers may run into, rounded up
__device__ __forceinline__ uint64_t dec_tail_start(const DecV1Args &V)
{
    const uint64_t fw = V.d.offsets[V.d.numPackets] >> 2;
    return fw > 16 ? fw - 16 : 0;
}
// word i of the stream, MSB first
__device__ __forceinline__ uint32_t dec_word(const DecV1Args &V, uint64_t tailStart, uint64_t i)
{
    if (!V.raw) return V.words[i];
    const uint64_t t = min(i - tailStart, (uint64_t)(kDecTailWords - 1));
    return __builtin_bswap32(i < tailStart ? V.raw[i] : V.tail[t]);
}
// 16 consecutive words from index w on (w + 16 <= the word limit): DIRECT windows that start in front of tailStart end in front
// of tailStart + 16 = the stream's last whole dword
__device__ __forceinline__ const uint32_t *dec_window(const DecV1Args &V, uint64_t tailStart, uint64_t w)
{
    if (!V.raw) return V.words + w;
    return w < tailStart ? V.raw + w : V.tail + min(w - tailStart, (uint64_t)(kDecTailWords - 16));
}

// the same with the source known at compile time (the entropy lanes and k_dec_raw: the fused launch, whose entropy wave is the
// launch's serial chain, must not pay a select per word for a mode it never uses)
template <bool DIRECT>
__device__ __forceinline__ uint32_t dec_word_t(const DecV1Args &V, uint64_t tailStart, uint64_t i)
{
    if constexpr (!DIRECT) return V.words[i];
    const uint64_t t = min(i - tailStart, (uint64_t)(kDecTailWords - 1));
    return __builtin_bswap32(i < tailStart ? V.raw[i] : V.tail[t]);
}
template <bool DIRECT>
__device__ __forceinline__ const uint32_t *dec_window_t(const DecV1Args &V, uint64_t tailStart, uint64_t w)
{
    if constexpr (!DIRECT) return V.words + w;
    return w < tailStart ? V.raw + w : V.tail + min(w - tailStart, (uint64_t)(kDecTailWords - 16));
}

// k_dec_tail (DIRECT): the end of the stream as dwords in the caller's byte order, zero-filled behind the last byte
__global__ __launch_bounds__(kDecTailWords) void k_dec_tail(const uint8_t *stream, const uint64_t *offsets, uint32_t numPackets, uint32_t *tail,
                                                            DecZero zero)
{
    dec_zero(zero);
    const uint64_t total = offsets[numPackets];
    const uint64_t fw = total >> 2, ts = fw > 16 ? fw - 16 : 0;
    const uint64_t i = ts + threadIdx.x;
    uint32_t v = 0;
    if ((i + 1) * 4 <= total) {
        v = ((const uint32_t *)stream)[i];
    } else {
        for (uint32_t b = 0; b < 4; b++)
            if (i * 4 + b < total) v |= (uint32_t)stream[i * 4 + b] << (8 * b);
    }
    tail[threadIdx.x] = v;
}

// Work lists of the separate-launch regime, built by k_dec_header (one lane per packet) in the progress words, which that
// regime does not use:  [0, 2n) chains the one-lane predictor takes, 4-tap chains from the front, 8-tap chains from the back
// | 16 counters: c4, c8, pairs A, pairs B, raw, rest | [n] pairs (class A from the front, B from the back) | [n] packets with
// an uncompressed element (k_dec_raw) | [n] packets k_dec_unmix has to write (everything that is neither a pair nor an
// uncompressed 16-bit stereo element, whose kernels write the PCM themselves: DecRec::pad2).
constexpr uint32_t kDecCounters = 16;
struct DecLists {
    uint32_t *chains, *cnt, *pairs, *raw, *rest;
    uint32_t *any;   // [2n] chains of mode 0 with 1..8 taps and any denShift that are neither 4- nor 8-tap / denShift-9 chains:
                     // what OTHER encoders emit (ffmpeg: orders 4..6, a shift per frame) — unpc_any_body, counter 6
};
__device__ __host__ inline DecLists dec_lists(const DecV1Args &V)
{
    const uint64_t n = V.d.numPackets;
    DecLists L;
    L.chains = V.prog;
    L.cnt = V.prog + 2 * n;
    L.pairs = L.cnt + kDecCounters;
    L.raw = L.pairs + n;
    L.rest = L.raw + n;
    L.any = L.rest + n;
    return L;
}
// Pair mode (16-bit stereo into a stereo frame, the shape of the benchmark and of most files): a packet whose two chains
// both take the fast predictor path is listed as a PAIR — class A (4 + 4 taps), class B (anything with an 8-tap chain) —
// and its chains sit in adjacent lanes of unpc_pair_body, which un-mixes and writes the PCM itself.
__device__ __host__ inline bool dec_stereo16(const DecV1Args &V)
{
    return V.d.bitDepth == 16 && V.d.numChannels == 2 && V.outChannels == 2 && V.elemBit == nullptr;
}

// lane states of the entropy kernel
enum : uint32_t { kStIdle = 0, kStGolomb = 1, kStRaw = 2 };

// ---- k_dec_header: element / header parse, once per packet, with the plain byte reader
// (codec/ALACDecoder.cu:600-700).  Leaves the record, the payload position and the status behind.
constexpr int kHdrWaves = 4;  // waves per workgroup of k_dec_header (the work-list counters are bumped once per workgroup)
__global__ __launch_bounds__(64 * kHdrWaves) void k_dec_header(DecV1Args V)
{
    const DecodeArgs &A = V.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t p = blockIdx.x * (64u * kHdrWaves) + tid;
    const bool live = p < A.numPackets;
    const uint64_t off = live ? A.offsets[p] : 0;
    const uint64_t nbytes = live ? A.offsets[p + 1] - off : 0;
    const uint8_t *base = A.stream + off;
    DecRec *rec = A.recs + (live ? p : 0);

    uint64_t hpos = (live && V.elemBit) ? V.elemBit[p] : 0;
    // The first 64 bytes of every packet (a compressed stereo element's header with two 8-tap coefficient sets is 45) as
    // seventeen MSB-first words in LDS, bytes behind the packet's end zeroed as the byte reader reads them: a field is then two
    // LDS words and a funnel shift instead of five byte loads, each a round trip of its own for 64 lanes in 64 different lines
    // (the launch took 0.114 ms of the 5.1 ms decode pass at 125 000 packets for 3 M wave-instructions).  Fields outside the
    // window, packets within 80 bytes of the stream's end and streams that are not dword aligned use the byte reader.
    __shared__ uint32_t win[64 * kHdrWaves * kHdrWinStride];
    const uint64_t totalBytes = A.offsets[A.numPackets];
    const bool useWin = live && ((uintptr_t)A.stream & 3) == 0 && (off & ~3ull) + 4 * kHdrWinWords <= totalBytes;
    const uint32_t phase = (uint32_t)(off & 3) * 8;  // bit of the packet's first byte inside window word 0
    if (useWin) {
        const uint32_t *sw = (const uint32_t *)(A.stream + (off & ~3ull));
        const int64_t endRel = (int64_t)nbytes + (int64_t)(off & 3);  // first byte (window-relative) behind the packet
#pragma unroll
        for (int i = 0; i < kHdrWinWords; i++) {
            uint32_t w = __builtin_bswap32(sw[i]);
            const int64_t keep = endRel - 4 * i;  // bytes of this word that belong to the packet
            if (keep <= 0) w = 0;
            else if (keep < 4) w &= 0xffffffffu << (8 * (4 - (int)keep));
            win[tid * kHdrWinStride + i] = w;
        }
    }
    auto rd = [&](uint32_t n) -> uint32_t {
        const uint64_t bit = hpos + phase;
        if (useWin && bit + 32 <= 32ull * (kHdrWinWords - 1)) {
            const uint32_t wi = (uint32_t)(bit >> 5), sh = (uint32_t)(bit & 31);
            const uint64_t two = ((uint64_t)win[tid * kHdrWinStride + wi] << 32) | win[tid * kHdrWinStride + wi + 1];
            hpos += n;
            return n ? (uint32_t)((two << sh) >> 32) >> (32 - n) : 0u;
        }
        return read_bits(base, nbytes, hpos, n);
    };
    uint32_t numSamples = A.frameSize, ech = 0, shb = 0, chanBits = 0, esc = 0;
    int32_t status = (live && V.round > 0) ? A.statusOut[p] : 0;  // a packet that failed in an earlier round stays failed
    uint32_t pbU = A.pb, pbV = A.pb;
    bool okc[2] = {false, false}, widec[2] = {false, false};  // chain c passes unpc_fast_ok's header part / has 8 taps
    bool anyc[2] = {false, false};                            // ... or unpc_any_ok's: mode 0, 1..8 taps, denShift 1..15
    DecRec R;
    R.numSamples = 0;
    R.escape = 0;
    R.mixBits = R.mixRes = 0;
    R.bytesShifted = 0;
    R.elementChannels = 0;
    R.shiftPos = 0;
    bool haveElement = false;
    if (live) {
        bool done = false;
        while (!done && status == 0) {
            if (!((hpos >> 3) < nbytes)) {  // :615
                status = -50;
                break;
            }
            const uint32_t tag = rd(3);
            switch (tag) {
            case 0:    // ID_SCE
            case 3:    // ID_LFE
            case 1: {  // ID_CPE
                ech = (tag == 1) ? 2u : 1u;
                if (ech != A.numChannels) {  // > 2-channel layouts (several elements) are not built yet
                    status = -4;
                    break;
                }
                (void)rd(4);
                if (rd(12) != 0) {  // :633 / :768
                    status = -50;
                    break;
                }
                const uint32_t hb = rd(4);
                const uint32_t partial = hb >> 3;
                shb = (hb >> 1) & 3;
                esc = hb & 1;
                if (shb == 3) {
                    status = -50;
                    break;
                }
                chanBits = A.bitDepth - shb * 8 + (ech == 2 ? 1 : 0);
                if (partial) numSamples = rd(32);
                if (numSamples > A.frameSize) {
                    status = -50;
                    break;
                }
                R.elementChannels = ech;
                if (!esc) {
                    R.mixBits = (int32_t)rd(8);
                    R.mixRes = (int8_t)rd(8);
                    for (uint32_t c = 0; c < ech; c++) {
                        uint32_t b = rd(8);
                        rec->c[c].mode = (uint16_t)(b >> 4);
                        rec->c[c].denShift = (uint16_t)(b & 0xf);
                        b = rd(8);
                        rec->c[c].pbFactor = (uint16_t)(b >> 5);
                        rec->c[c].num = (uint16_t)(b & 0x1f);
                        okc[c] = rec->c[c].mode == 0 && rec->c[c].denShift == kDenShift && ((b & 0x1f) == 4 || (b & 0x1f) == 8);
                        widec[c] = (b & 0x1f) == 8;
                        anyc[c] = !okc[c] && rec->c[c].mode == 0 && (b & 0x1f) >= 1 && (b & 0x1f) <= 8 && rec->c[c].denShift >= 1;
                        if (c == 0) pbU = (A.pb * (b >> 5)) / 4;  // :825
                        else pbV = (A.pb * (b >> 5)) / 4;        // :841
                        for (uint32_t i = 0; i < (b & 0x1f); i++)
                            rec->c[c].coefs[i] = (int16_t)rd(16);
                    }
                    R.shiftPos = hpos;
                    if (shb) hpos += (uint64_t)shb * 8 * ech * numSamples;
                    R.bytesShifted = shb;
                } else {
                    chanBits = A.bitDepth;  // :697-727 / :856-896 uncompressed element
                    R.escape = 1;
                }
                R.numSamples = numSamples;
                haveElement = true;
                done = true;  // channelIndex >= numChannels, :967
                break;
            }
            case 2:  // ID_CCE
            case 5:  // ID_PCE
                status = -50;
                break;
            case 4: {  // ID_DSE :1033-1059
                (void)rd(4);
                const uint32_t align = rd(1);
                uint32_t count = rd(8);
                if (count == 255) count += rd(8);
                if (align && (hpos & 7)) hpos += 8 - (hpos & 7);
                hpos += (uint64_t)count * 8;
                if ((hpos + 7) / 8 > nbytes) status = -50;
                break;
            }
            case 6: {  // ID_FIL :1012-1027
                int32_t count = (int32_t)rd(4);
                if (count == 15) count += (int32_t)rd(8) - 1;
                hpos += (uint64_t)count * 8;
                if ((hpos + 7) / 8 > nbytes) status = -50;
                break;
            }
            default:  // ID_END before any audio element
                done = true;
                break;
            }
        }
    }

    if (live && status == 0) {
        // Untrusted input: (1) every byte of the packet must be inside the staged word stream — a stream longer than the
        // workspace was sized for (legal ID_FIL / ID_DSE padding can exceed alac_hip_decode_workspace_bytes' bound) is
        // truncated by k_dec_stage, so its tail packets fail here instead of decoding bytes that are not theirs;
        // (2) an uncompressed element's fixed-width payload must end inside the packet (the reference's overrun test,
        // codec/ALACDecoder.cu:996-1000, returns kALAC_ParamError for the same packet).
        const uint64_t stagedBytes = (V.capWords - 64) * 4;  // k_dec_stage keeps 64 zero words behind what it copies
        if (!V.raw && off + nbytes > stagedBytes) status = -50;  // (DIRECT reads the caller's stream: nothing is truncated)
        if (haveElement && R.escape && hpos + (uint64_t)R.numSamples * R.elementChannels * A.bitDepth > nbytes * 8) status = -50;
    }
    if (live) {
        rec->numSamples = R.numSamples;
        rec->escape = R.escape;
        rec->mixBits = R.mixBits;
        rec->mixRes = R.mixRes;
        rec->bytesShifted = R.bytesShifted;
        rec->elementChannels = haveElement ? R.elementChannels : 0;
        rec->shiftPos = R.shiftPos;
        rec->status = status;
        rec->pad = (uint32_t)hpos;  // first payload bit (entropy coded or raw), from the packet start
        A.statusOut[p] = status;
        if (V.mismatch && status == -4) atomicAdd(V.mismatch, 1u);  // gates the lane decoder behind this pipeline
        if (V.round == 0 || haveElement || status != 0) A.numSamplesOut[p] = status == 0 ? R.numSamples : 0;
        // where the next element starts: known here for an uncompressed element, left by the entropy lane otherwise
        if (V.elemBit && status == 0 && haveElement && R.escape)
            V.elemBit[p] = (uint32_t)(hpos + (uint64_t)R.numSamples * R.elementChannels * A.bitDepth);
    }
    // who writes this packet's PCM: the predictor lanes of a pair or k_dec_raw (pad2 = 1), or k_dec_unmix
    const bool good = live && status == 0;
    const bool stereo16 = dec_stereo16(V);
    // unpc_fast_ok; chanBits <= 23: the fast predictor bodies multiply coefficient x (top - sample) with 24-bit multiplies,
    // and a difference of two chanBits-wide samples has chanBits + 1 significant bits (24-bit material coded without
    // shift-off bytes, 32-bit mono: the generic predictor's job)
    const bool fastShape = (A.frameSize & 7) == 0 && R.numSamples >= 16 && haveElement && !R.escape && chanBits <= kFastChanBits;
    const bool ok0 = good && fastShape && okc[0], ok1 = good && fastShape && R.elementChannels == 2 && okc[1];
    // pairs: 16-bit stereo, and (round 4) 20- / 24-bit stereo with at most one shifted-off byte per sample — what every encoder
    // emits for that material (unpc_pair_body<T, DEPTH> re-attaches the bytes and packs the 3-byte samples itself)
    const bool pairDepth = stereo16 || ((A.bitDepth == 24 || A.bitDepth == 20) && A.numChannels == 2 && V.outChannels == 2 &&
                                        V.elemBit == nullptr && R.bytesShifted <= 1);
    const bool pair = pairDepth && V.lists && V.pairs && ok0 && ok1;
    const bool rawP = good && haveElement && R.escape != 0;
    const bool rawDirect = rawP && stereo16 && R.elementChannels == 2;
    if (live) rec->pad2 = (pair || rawDirect) ? 1u : 0u;
    if (V.lists) {
        // Every entry of a list costs its wave a fetch-and-add on the list's counter, and 125 000 packets are 1 954 waves bumping
        // the same three or four words: ~50 of the launch's 90 us were that queue at the L2.  The waves of a workgroup first
        // add up in LDS, one lane per counter then bumps the global word once for the whole workgroup (a quarter of the
        // atomics, issued side by side instead of one round trip after the other), and the entries are written behind a barrier.
        const DecLists L = dec_lists(V);
        const uint64_t below = (1ull << lane) - 1;
        __shared__ uint32_t blkCnt[8], blkBase[8];
        if (tid < 8) blkCnt[tid] = 0;
        __syncthreads();
        constexpr int kPushes = 10;
        uint32_t slot[kPushes];  // this lane's index inside its workgroup's share of the counter, or ~0u
        int np = 0;
        auto count = [&](bool mine, uint32_t counter) {
            const uint64_t m = __ballot(mine);
            uint32_t b = 0;
            if (lane == 0 && m) b = atomicAdd(&blkCnt[counter], (uint32_t)__popcll(m));  // LDS
            b = (uint32_t)__shfl((int)b, 0) + (uint32_t)__popcll(m & below);
            slot[np++] = mine ? b : ~0u;
        };
        const uint32_t total = A.numPackets * A.numChannels;
        const bool wide0 = widec[0], wide1 = widec[1];
        count(ok0 && !pair && !wide0, 0);
        count(ok0 && !pair && wide0, 1);
        count(ok1 && !pair && !wide1, 0);
        count(ok1 && !pair && wide1, 1);
        count(pair && !wide0 && !wide1, 2);
        count(pair && (wide0 || wide1), 3);
        count(good && fastShape && anyc[0], 6);
        count(good && fastShape && R.elementChannels == 2 && anyc[1], 6);
        count(rawP, 4);
        count(good && !pair && !rawDirect, 5);
        __syncthreads();
        if (tid < 8 && blkCnt[tid]) blkBase[tid] = atomicAdd(L.cnt + tid, blkCnt[tid]);
        __syncthreads();
        np = 0;
        auto place = [&](uint32_t counter, uint32_t *list, uint32_t value, bool fromBack, uint32_t size) {
            const uint32_t b = slot[np++];
            if (b != ~0u) {
                const uint32_t at = blkBase[counter] + b;
                list[fromBack ? size - 1 - at : at] = value;
            }
        };
        place(0, L.chains, p * A.numChannels, false, total);
        place(1, L.chains, p * A.numChannels, true, total);
        place(0, L.chains, p * A.numChannels + 1, false, total);
        place(1, L.chains, p * A.numChannels + 1, true, total);
        place(2, L.pairs, p, false, A.numPackets);
        place(3, L.pairs, p, true, A.numPackets);
        place(6, L.any, p * A.numChannels, false, total);
        place(6, L.any, p * A.numChannels + 1, false, total);
        place(4, L.raw, p, false, A.numPackets);
        place(5, L.rest, p, false, A.numPackets);
    }
    (void)pbU;
    (void)pbV;
}

// ---- k_dec_raw: uncompressed (escape) elements are fixed-width fields, i.e. not serial at all: one thread per
// sample-frame reads its fields straight from the staged words (codec/ALACDecoder.cu:697-727 / :856-896)
template <bool DIRECT>
__device__ __forceinline__ void raw_body(const DecV1Args &V, uint32_t p, uint32_t first, uint32_t step)
{
    const DecodeArgs &A = V.d;
    const DecRec *rec = A.recs + p;
    if (rec->status != 0 || !rec->escape || rec->elementChannels == 0) return;
    const uint32_t ech = rec->elementChannels, n = rec->numSamples, w = A.bitDepth;
    const uint64_t off = A.offsets[p];
    const uint64_t bitBase = (off & 3) * 8 + rec->pad;
    const uint64_t wbase = off >> 2, tailStart = DIRECT ? dec_tail_start(V) : 0;
    const uint64_t lastWord = V.capWords - 2 - (off >> 2);  // k_dec_header has checked the payload; never index past the stage
    int32_t *rowU = V.plane + (uint64_t)p * A.numChannels * A.frameSize;
    // pad2 (k_dec_header): a 16-bit stereo element into a stereo frame — the fields ARE the PCM (codec/ALACDecoder.cu:856-874),
    // written here as one word per frame instead of going through the plane and k_dec_unmix
    const bool direct = rec->pad2 != 0;
    uint32_t *pcm = (uint32_t *)(A.pcmOut + (uint64_t)p * A.frameSize * 4);
    // four sample-frames per round, every load of the round issued before its first store: the loop is bound by the
    // chain of dependent round trips per wave, not by bytes (an iteration per frame took 1.24 ms for the 15 600 escape
    // packets of the 125 000-packet benchmark)
    constexpr uint32_t U = 4;
    uint32_t done = 0;  // frames [0, done) are written by the group path below
    if (direct && ech == 2 && w == 16) {
        // The benchmark's escape packets (one in eight): frame j IS the 32 bits from bit bitBase + 32 j on — L then R, MSB first —
        // so four frames are five consecutive words at one bit phase: a 16-byte load and a dword, four funnel shifts, four
        // half-word swaps (the PCM word is L | R << 16), one 16-byte store.  The per-field code below spent ~100 instructions
        // per frame on them (k_dec_raw 0.20 ms of the 5.1 ms decode pass at 125 000 packets, 87 M wave-instructions).
        typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t s = (uint32_t)(bitBase & 31);
        const uint64_t a0 = wbase + (bitBase >> 5);
        // groups whose five words all come from the stream itself (DIRECT: in front of the tail copy; staged: inside the stage)
        const uint64_t wordEnd = DIRECT ? tailStart : V.capWords - 2;
        const uint32_t groups = n / 4;
        uint32_t safe = 0;
        if (a0 + 5 <= wordEnd) safe = (uint32_t)min((uint64_t)groups, (wordEnd - a0 - 5) / 4 + 1);
        const uint32_t *src = DIRECT ? V.raw : V.words;
        for (uint32_t g = first; g < safe; g += step) {
            const U4 q = *(const U4 *)(src + a0 + 4 * (uint64_t)g);
            const uint32_t e = src[a0 + 4 * (uint64_t)g + 4];
            uint32_t x[5] = {q.x, q.y, q.z, q.w, e};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 5; k++) x[k] = DIRECT ? __builtin_bswap32(x[k]) : x[k];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t v = (uint32_t)(((((uint64_t)x[k] << 32) | x[k + 1]) << s) >> 32);
                o[k] = (v >> 16) | (v << 16);
            }
            const U4 t4 = {o[0], o[1], o[2], o[3]};
            *(U4 *)(pcm + 4 * (uint64_t)g) = t4;
        }
        done = safe * 4;
    }
    for (uint32_t j0 = done + first; j0 < n; j0 += U * step) {
        uint32_t hi[U][2], lo[U][2], shs[U][2];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t j = j0 + u * step;
#pragma unroll
            for (uint32_t c = 0; c < 2; c++) {
                const uint64_t b = bitBase + ((uint64_t)(j < n ? j : 0) * ech + (c < ech ? c : 0)) * w;
                const uint32_t i = (uint32_t)min((uint64_t)(b >> 5), lastWord);
                shs[u][c] = (uint32_t)(b & 31);
                hi[u][c] = dec_word_t<DIRECT>(V, tailStart, wbase + i);
                lo[u][c] = dec_word_t<DIRECT>(V, tailStart, wbase + i + 1);
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t j = j0 + u * step;
            uint32_t f[2] = {0, 0};
#pragma unroll
            for (uint32_t c = 0; c < 2; c++) {
                if (j < n && c < ech) {
                    const uint64_t two = ((uint64_t)hi[u][c] << 32) | lo[u][c];
                    const uint32_t v = (uint32_t)((two << shs[u][c]) >> 32) >> (32 - w);
                    f[c] = v;
                    if (!direct) (rowU + c * A.frameSize)[j] = (int32_t)(v << (32 - w)) >> (32 - w);
                }
            }
            if (direct && j < n) pcm[j] = f[0] | (f[1] << 16);
        }
    }
}

// blocks per packet of the sample-parallel kernels (1-D grids: gridDim.y stops at 65535 packets)
__device__ __host__ inline uint32_t blocks_per_packet(uint32_t frameSize) { return frameSize > 1024 ? (frameSize + 1023) / 1024 : 1; }

// A workgroup walks packets blockIdx.x, blockIdx.x + gridDim.x, ...: most packets are not escapes and a workgroup per
// packet (times the blocks of a frame) spent the launch on workgroups that read one record and left — 1.23 ms for
// 500 000 workgroups at 125 000 packets, 15 600 of them with work.
// (round 3: the packets come from k_dec_header's list of uncompressed elements, so nobody reads records to find them)
__global__ __launch_bounds__(256) void k_dec_raw(DecV1Args V)
{
    const DecLists L = dec_lists(V);
    const uint32_t count = L.cnt[4];
    if (V.raw) {  // wave-uniform
        for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) raw_body<true>(V, L.raw[i], threadIdx.x, blockDim.x);
    } else {
        for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) raw_body<false>(V, L.raw[i], threadIdx.x, blockDim.x);
    }
}

// per-lane state of the entropy kernel
struct EntLane {
    BitWin bw;
    uint32_t F;            // first word (relative to the packet's first staged word) not yet in the ring
    uint32_t active;       // still decoding
    uint32_t chan, c, mb, zmode, pb;
    uint32_t endPos;       // bits from the packet start to the end of the element's last channel
    int32_t status;
    int32_t *row;          // row of the channel being decoded
};

// The rounds of one wave.  PB40: every lane uses pb = 40 (pbFactor 4 with the standard cookie — every stream this
// library or Apple's encoder writes), so pb * x is two shifts; otherwise full 32-bit multiplies.
// PUB: fused launch — the lane publishes how many residuals are complete every (pubMask + 1) rounds (drain,
// agent-scope release, one flag store per channel) for the predictor waves that follow it.
// WIDE (separate launches only): the residual stores of a round are DEFERRED by one round.  gfx9 counts loads and stores
// in one counter (vmcnt), so the wait for the staged words a round fetched ahead also waited for the 4-byte stores the
// symbols in between had just issued — a full write round trip under load, every 16 symbols (entropy kernel at 125 000
// packets: 4.1 ms with the stores, 2.8 ms without them).  Here a round only RECORDS (position, value) per symbol; the
// stores are issued at the start of the next round, right after the wait, and have a whole round to complete.  A symbol
// that was not decoded repeats the lane's previous pair (same value to the same address: harmless).
template <bool PB40, bool PUB, bool WIDE = false, bool ZFILL = WIDE, bool LOCAL = false, bool DIRECT = false>
__device__ __forceinline__ void entropy_rounds(EntLane &E, const DecV1Args &V, uint32_t *ringRow, uint64_t wordBase,
                                               uint32_t cur0, uint32_t bit0, uint32_t limit, uint64_t nbytes,
                                               uint32_t numSamples, uint32_t ech, uint32_t chanBits, uint32_t pbV,
                                               uint32_t *prog)
{
    const DecodeArgs &A = V.d;
    const uint32_t wb = (1u << A.kb) - 1;
    const uint64_t wordLimit = min(V.capWords, ((A.offsets[A.numPackets] + 3) >> 2) + 64);  // what k_dec_stage wrote
    const uint64_t tailStart = DIRECT ? dec_tail_start(V) : 0;
    constexpr bool swap = DIRECT;  // the caller's byte order
    auto fetch16 = [&](uint32_t rel, uint32_t (&q)[16]) {
        uint64_t w = wordBase + rel;
        const uint64_t lastStart = wordLimit - kWinWords;  // corrupt input may run past the packet: stay inside
        w = w < lastStart ? w : lastStart;
        // four 16-byte loads (the staged words are dword aligned only): a quarter of the L2 requests of sixteen dword loads
        typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
        const U4 *sw = (const U4 *)dec_window_t<DIRECT>(V, tailStart, w);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const U4 t = sw[i];
            q[4 * i] = swap ? __builtin_bswap32(t.x) : t.x;
            q[4 * i + 1] = swap ? __builtin_bswap32(t.y) : t.y;
            q[4 * i + 2] = swap ? __builtin_bswap32(t.z) : t.z;
            q[4 * i + 3] = swap ? __builtin_bswap32(t.w) : t.w;
        }
    };
    auto write16 = [&](uint32_t rel, const uint32_t (&q)[16]) {
        uint32_t *dst = ringRow + (rel & 31u);
#pragma unroll
        for (int i = 0; i < 16; i++) dst[i] = q[i];
    };
    auto mul_pb = [&](uint32_t x) { return PB40 ? times5(x) << 3 : E.pb * x; };  // wraps like the reference's uint32

    uint32_t q[16];
    bool pending = false;
    uint32_t pendBase = 0;
    uint32_t round = 0;
    BitWin &bw = E.bw;
    auto publish = [&]() {
        if constexpr (PUB) {
            // per-lane 4-byte stores cannot be written through one by one (16 x HBM write amplification, measured):
            // plain stores, and an agent-scope release (L2 write-back, ~10 us) at every publish instead
            const uint32_t all = 0xffffffffu;
            const uint32_t u = (!E.active || E.chan > 0) ? all : E.c;
            const uint32_t v = !E.active ? all : (E.chan > 0 ? E.c : 0u);
            if constexpr (LOCAL) {
                // the followers are waves of THIS workgroup (k_dec_fused_wg): same CU, same L1 and L2 — the stores only
                // have to have left the wave (workgroup-scope release: no cache write-back), and the progress words are LDS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (prog && !A.ho.lose) {
                    __hip_atomic_store(prog, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(prog + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (prog && !A.ho.lose) {
                    __hip_atomic_store(prog, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(prog + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    // WIDE: the pairs recorded by the previous round, as int32 indices from the lane's first row
    uint32_t dpos[kDecRound];
    int32_t dval[kDecRound];
    uint32_t lastPos = 0;   // a coded lane's first row starts as zeros (pre-zeroed plane): (0, 0) is a harmless first pair
    int32_t lastVal = 0;
    uint32_t rowOff = 0;    // E.row - rowBase, in samples
    int32_t *const rowBase = E.row;
    const bool storer = E.active != 0;  // lanes without a coded packet never store (an escape packet's rows are k_dec_raw's)
#pragma unroll
    for (int k = 0; k < kDecRound; k++) {
        dpos[k] = 0;
        dval[k] = 0;
    }
    uint32_t lim = 0;   // see the round loop
    bool had = false;   // this lane decoded something in the round the pairs come from
    bool full = false;  // ... and decoded its last symbol (so all of them)
    // A lane that stops decoding inside a round stays stopped (and repeats its last pair), and positions only grow: when the
    // LAST symbol of the round was decoded all sixteen were, and "last - first == 15" then means sixteen CONSECUTIVE
    // residuals — every round of a lane outside zero runs.  Those leave as four 16-byte stores instead of sixteen 4-byte ones: the 64 lanes of a store instruction hit 64
    // different cache lines either way, and it is the number of such instructions that the CU's one address path serialises.
    auto store_deferred = [&]() {
        if constexpr (WIDE) {
            typedef int32_t I4 __attribute__((ext_vector_type(4), aligned(4)));
            const bool mine = storer && had;
            const bool run16 = mine && full && dpos[kDecRound - 1] - dpos[0] == (uint32_t)(kDecRound - 1);
            if (run16) {
                int32_t *dst = rowBase + dpos[0];
#pragma unroll
                for (int k = 0; k < kDecRound; k += 4) {
                    I4 t = {dval[k], dval[k + 1], dval[k + 2], dval[k + 3]};
                    *(I4 *)(dst + k) = t;
                }
            }
            if (__any(mine && !run16)) {
                if (mine && !run16) {
#pragma unroll
                    for (int k = 0; k < kDecRound; k++) rowBase[dpos[k]] = dval[k];
                }
            }
            had = false;
            full = false;
        }
    };
    while (__any(E.active != 0)) {
        if (pending) {
            write16(pendBase, q);
            E.F = pendBase + 16;
            pending = false;
        }
        store_deferred();  // behind the wait above, in front of the next fetch
        if (E.active && E.F - (bw.idx - 2u) <= 16) {
            fetch16(E.F, q);
            pending = true;
            pendBase = E.F;
        }
        asm volatile("" ::: "memory");
        // a lane decodes while it is active and has >= 128 staged bits ahead of its window: F only changes between rounds,
        // so that is "idx <= lim" with one limit per round (0 = never: idx starts at 2); whoever clears `active` clears it
        lim = E.active ? E.F - 4u : 0u;
#pragma unroll
        for (int it = 0; it < kDecRound; it++) {
            if constexpr (WIDE) {
                dpos[it] = lastPos;
                dval[it] = lastVal;
            }
            if (bw.idx <= lim) {
                // ---- one residual: dyn_get_32bit (ag_dec.c:220-270), straight-line for the common case ----
                const uint32_t k = min(22u - (uint32_t)__builtin_clz(E.mb + (3u << kQBShift)), A.kb);  // lg3a(mb >> 9)
                const uint32_t stream = bw_peek(bw);
                const uint32_t pre = (uint32_t)__builtin_clz(~stream | 1u);  // leading ones, at most 31
                // the k bits behind the prefix and its closing zero (only meaningful below the escape: pre <= 8, k <= 16)
                const uint32_t v = __builtin_amdgcn_ubfe(stream, 31u - k - pre, k);
                const uint32_t big = v >= 2 ? 1u : 0u;
                // ag_dec.c:248-262 without its k != 1 case: with k = 1, m = 1 and v < 2, so the general form gives the
                // same value (pre) and the same length (pre + 1)
                uint32_t n = (pre << k) - pre + (big ? v - 1 : 0u);
                uint32_t used = pre + k + big;
                // (ag_dec.c:302 stops a channel whose next residual would start at or past the packet's end; such a
                // channel also fails the end check below — the position only grows — so the status is the same and
                // the per-residual test is not repeated here)
                if (pre >= kMaxPrefix) {
                    bw_skip(bw, kMaxPrefix);
                    n = bw_get(bw, chanBits);
                    used = 0;
                }
                bw_skip(bw, used);
                const uint32_t nd = n + E.zmode;
                // ((nd + 1) >> 1) * (nd odd ? -1 : 1); stored even when the packet just failed (c < numSamples still)
                const int32_t val = (int32_t)((nd >> 1) ^ (0u - (nd & 1u)));
                if constexpr (!WIDE) {
                    E.row[E.c] = val;
                } else {
                    lastPos = rowOff + E.c;
                    lastVal = val;
                    had = true;
                    full = it == kDecRound - 1;
                    dpos[it] = lastPos;
                    dval[it] = val;
                }
                E.c++;
                uint32_t mb = mul_pb(nd) + E.mb - (mul_pb(E.mb) >> kQBShift);
                mb = n > kMeanClamp ? kMeanClamp : mb;
                E.mb = mb;
                E.zmode = 0;
                // (a lane in here is active: lim is 0 otherwise)
                const bool low = (mb << 2) < (1u << kQBShift), atEnd = E.c >= numSamples;
                if (__any(low | atEnd)) {
                    const bool zrun = low && !atEnd;
                    int32_t *zfAt = nullptr;  // ZFILL: where this lane's run of zfCnt zeros starts
                    uint32_t zfCnt = 0;
                    if (zrun) {
                        // zero run (ag_dec.c:324-352): only the index moves (the plane is pre-zeroed, or — ZFILL — the zeros
                        // are written right behind this block)
                        E.zmode = 1;
                        const uint32_t kz = (uint32_t)(lead(mb) - 24 + (int32_t)((mb + 16u) >> 6));
                        const uint32_t mz = ((1u << kz) - 1) & wb;
                        const uint32_t st2 = bw_peek(bw);
                        const uint32_t prz = (uint32_t)__builtin_clz(~st2 | 1u);
                        uint32_t nz;
                        if (prz >= kMaxPrefix) {
                            nz = (st2 << kMaxPrefix) >> (32 - kMaxRunBits);
                            bw_skip(bw, kMaxPrefix + kMaxRunBits);
                        } else {
                            const uint32_t vz = (st2 << (prz + 1)) >> (32 - kz);
                            nz = prz * mz + vz - 1;
                            uint32_t uz = prz + 1 + kz;
                            if (vz < 2) {
                                nz -= (vz - 1);
                                uz -= 1;
                            }
                            bw_skip(bw, uz);
                        }
                        if (!((uint64_t)E.c + nz <= (uint64_t)numSamples)) {  // :341
                            E.status = -50;
                            E.active = 0;
                            lim = 0;
                        } else if constexpr (ZFILL) {
                            zfAt = rowBase + rowOff + E.c;
                            zfCnt = nz;
                        }
                        E.c += nz;
                        if (nz >= 65535) E.zmode = 0;
                        E.mb = 0;
                    }
                    if constexpr (ZFILL) {
                        // This launch does not clear the plane first (a 4.1 GB fill at 125 000 stereo packets):
                        // the zeros of a run are written here, by the WHOLE wave for one lane's run at a time — 1 KB per store
                        // instruction (lane-by-lane, a silent channel alone was 1024 serial 16-byte stores of one lane).
                        typedef int32_t I4 __attribute__((ext_vector_type(4), aligned(4)));
                        const I4 zero4 = {0, 0, 0, 0};
                        // (this is inside the per-lane "decode a symbol" branch: only the lanes still decoding are here)
                        const uint64_t here = __ballot(1);
                        const uint32_t nHere = (uint32_t)__popcll(here);
                        const uint32_t rank = (uint32_t)__popcll(here & ((1ull << (threadIdx.x & 63)) - 1));
                        for (uint64_t m = __ballot(zfCnt != 0); m; m &= m - 1) {
                            const int src = __builtin_ctzll(m);
                            const uint64_t at = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)zfAt >> 32), src) << 32) |
                                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)zfAt, src);
                            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)zfCnt, src);
                            int32_t *z = (int32_t *)at;
                            for (uint32_t i = rank * 4; i < cnt; i += nHere * 4) {
                                if (i + 4 <= cnt) {
                                    *(I4 *)(z + i) = zero4;
                                } else {
                                    for (uint32_t q = i; q < cnt; q++) z[q] = 0;
                                }
                            }
                        }
                    }
                    if (E.active && E.c >= numSamples) {
                        // channel complete: ag_dec.c:359 end check; the next channel starts where this one ended
                        if ((uint64_t)(bw_pos(bw) - bit0 + 7) / 8 > nbytes) {
                            E.status = -50;
                            E.active = 0;
                            lim = 0;
                        } else if (E.chan + 1 < ech) {
                            E.chan++;
                            E.c = 0;
                            E.mb = A.mb;
                            E.zmode = 0;
                            E.pb = pbV;
                            E.row += A.frameSize;
                            rowOff += A.frameSize;
                        } else {
                            E.endPos = bw_pos(bw) - bit0;
                            E.active = 0;
                            lim = 0;
                        }
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
        if (PUB && (++round & V.pubMask) == 0) publish();
    }
    store_deferred();  // the last round's pairs
    publish();  // everything is complete (or failed): nobody waits for this lane any more
}

// PPW: packets per wave (lanes PPW .. 63 idle); progLds: LOCAL progress words of the workgroup, [PPW][2]
template <bool PUB, bool WIDE = false, bool ZFILL = WIDE, int PPW = 64, bool DIRECT = false>
__device__ __forceinline__ void entropy_body(const DecV1Args &V, uint32_t *ring, uint32_t block, uint32_t *progLds = nullptr)
{
    constexpr bool LOCAL = PPW != 64;
    const DecodeArgs &A = V.d;
    const int lane = threadIdx.x & 63;
    const uint32_t p = block * (uint32_t)PPW + lane;
    const bool live = lane < PPW && p < A.numPackets;
    const uint64_t off = live ? A.offsets[p] : 0;
    const uint64_t nbytes = live ? A.offsets[p + 1] - off : 0;
    DecRec *rec = A.recs + (live ? p : 0);
    const int32_t status0 = live ? rec->status : 0;
    const uint32_t ech = live ? rec->elementChannels : 0;
    const uint32_t numSamples = live ? rec->numSamples : 0;
    const uint32_t chanBits = A.bitDepth - (live ? rec->bytesShifted : 0) * 8 + (ech == 2 ? 1 : 0);
    const uint32_t pbU = live ? (A.pb * rec->c[0].pbFactor) / 4 : A.pb;  // :825
    const uint32_t pbV = live ? (A.pb * rec->c[1].pbFactor) / 4 : A.pb;  // :841
    const uint32_t hpos = live ? rec->pad : 0;
    const bool coded = live && status0 == 0 && ech != 0 && !rec->escape && numSamples > 0;

    // bit positions are relative to the packet's first staged word (the packet starts at bit 8 * (off & 3))
    const uint64_t wordBase = off >> 2;
    const uint32_t bit0 = (uint32_t)(off & 3) * 8;
    const uint32_t limit = bit0 + (uint32_t)nbytes * 8;  // maxPos of ag_dec.c:295
    const uint32_t pos0 = bit0 + hpos;
    const uint32_t cur0 = pos0 >> 5;

    // ---- circular ring of 32 staged words per lane; word t (relative to the packet's first staged word) lives in
    // slot (t - cur0) & 31.  16 words are fetched whenever 16 slots are free, one round (16 symbols) before they
    // are written into the ring, so the global latency is covered by the round's own work.
    uint32_t *ringRow = ring + lane * kWinStride;
    const uint64_t wordLimit = min(V.capWords, ((A.offsets[A.numPackets] + 3) >> 2) + 64);
    EntLane E;
    {
        uint64_t w = wordBase + cur0;
        const uint64_t lastStart = wordLimit - kWinWords;
        w = w < lastStart ? w : lastStart;
        const uint32_t *sw = dec_window_t<DIRECT>(V, DIRECT ? dec_tail_start(V) : 0, w);
#pragma unroll
        for (int i = 0; i < 16; i++) ringRow[i] = DIRECT ? __builtin_bswap32(sw[i]) : sw[i];
    }
    asm volatile("" ::: "memory");
    // from here on word indices are relative to the first word staged (cur0): ring slot = index & 31 with no subtraction in
    // the per-residual path; bit positions follow (bit0R may wrap below zero: only differences are used)
    const uint64_t wordBaseR = wordBase + cur0;
    const uint32_t bit0R = bit0 - cur0 * 32u;
    E.F = 16;
    {
        E.bw.ring = ringRow;
        E.bw.w0 = ringRow[0];
        E.bw.w1 = ringRow[1];
        E.bw.o = pos0 & 31;
        E.bw.idx = 2;
        E.bw.nw = ringRow[2];
    }
    E.active = coded ? 1u : 0u;
    E.chan = 0;
    E.c = 0;
    E.mb = A.mb;
    E.zmode = 0;
    E.pb = pbU;
    E.status = status0;
    E.endPos = 0;
    E.row = V.plane + (uint64_t)p * A.numChannels * A.frameSize;

    uint32_t *prog = (PUB && live) ? (LOCAL ? progLds + lane * 2 : V.prog + (uint64_t)p * 2) : nullptr;
    if (__all(!coded || (pbU == 40 && pbV == 40)))
        entropy_rounds<true, PUB, WIDE, ZFILL, LOCAL, DIRECT>(E, V, ringRow, wordBaseR, 0u, bit0R, limit, nbytes, numSamples, ech, chanBits, pbV, prog);
    else
        entropy_rounds<false, PUB, WIDE, ZFILL, LOCAL, DIRECT>(E, V, ringRow, wordBaseR, 0u, bit0R, limit, nbytes, numSamples, ech, chanBits, pbV, prog);

    if (live && E.status != status0) {
        rec->status = E.status;
        A.statusOut[p] = E.status;
        A.numSamplesOut[p] = 0;
    }
    if (coded && V.elemBit && E.status == 0) V.elemBit[p] = E.endPos;
}

// separate launches of a large batch: deferred residual stores (a round's sixteen residuals leave as four 16-byte stores)
// Four waves to a workgroup: a CU then takes the entropy waves four at a time, one per SIMD.  As single-wave workgroups
// (about 7.6 per CU at 125 000 packets) some SIMD of a CU ended up with three of them, and the launch is as slow as that SIMD.
constexpr int kEntWavesPerWg = 4;
template <bool DIRECT>
__global__ __launch_bounds__(64 * kEntWavesPerWg) void k_dec_entropy_wide(DecV1Args V, uint32_t nEnt)
{
    __shared__ uint32_t ring[kEntWavesPerWg][64 * kWinStride];
    const uint32_t slot = threadIdx.x >> 6, b = blockIdx.x * (uint32_t)kEntWavesPerWg + slot;
    if (b >= nEnt) return;
    if constexpr (DIRECT) entropy_body<false, true, true, 64, true>(V, ring[slot], b);
    else entropy_body<false, true>(V, ring[slot], b);
}

// ---- unpc_block (codec/dp_dec.c:55-381), in place over the chain's row ----
// Fast kernel: what every ALAC encoder in the wild emits for 16/24-bit material and all this library's encoder
// emits — mode 0, denShift 9, 4 or 8 taps — on the 2-lanes-per-chain mapping of alac_lms.hpp (32 chains per
// wave).  Everything else (first-order mode, other tap counts or shifts, tiny packets) goes to the lane-serial
// generic kernel.  The two kernels partition the chains with unpc_fast_ok.
__device__ __forceinline__ bool unpc_fast_ok(const DecodeArgs &A, const DecRec *rec, uint32_t ch)
{
    if (rec->status != 0 || rec->escape || rec->elementChannels == 0) return false;
    const DecChan &c = rec->c[ch];
    const uint32_t chanBits = A.bitDepth - rec->bytesShifted * 8 + (rec->elementChannels == 2 ? 1 : 0);
    return (A.frameSize & 7) == 0 && rec->numSamples >= 16 && c.mode == 0 && c.denShift == kDenShift &&
           (c.num == 4 || c.num == 8) && chanBits <= kFastChanBits;
}

// chains the separate launches hand to unpc_any_body (the same test as k_dec_header's anyc && fastShape)
__device__ __forceinline__ bool unpc_any_ok(const DecodeArgs &A, const DecRec *rec, uint32_t ch)
{
    if (rec->status != 0 || rec->escape || rec->elementChannels == 0) return false;
    const DecChan &c = rec->c[ch];
    const uint32_t chanBits = A.bitDepth - rec->bytesShifted * 8 + (rec->elementChannels == 2 ? 1 : 0);
    const bool fast = c.mode == 0 && c.denShift == kDenShift && (c.num == 4 || c.num == 8);
    return (A.frameSize & 7) == 0 && rec->numSamples >= 16 && chanBits <= kFastChanBits && !fast && c.mode == 0 && c.num >= 1 &&
           c.num <= 8 && c.denShift >= 1;
}

// the first 16 samples (warm-up positions + a few regular steps), lane-serial; leaves coefficients and outputs
template <int NA>
__device__ __forceinline__ void unpc_head16(const int32_t (&del)[16], int32_t (&out)[16], int32_t (&a8)[8],
                                            uint32_t chanshift)
{
    Lms<NA> s;
#pragma unroll
    for (int k = 0; k < NA; k++) s.a[k] = a8[k];
#pragma unroll
    for (int k = 0; k <= NA; k++) s.h[k] = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int32_t o;
        if (j == 0) {
            o = del[0];
            lms_push<NA>(s, o);
        } else if (j <= NA) {
            o = sext(del[j] + s.h[0], chanshift);
            lms_push<NA>(s, o);
        } else {
            o = lms_step_dec<NA>(s, del[j], chanshift, kDenShift);
        }
        out[j] = o;
    }
#pragma unroll
    for (int k = 0; k < NA; k++) a8[k] = s.a[k];
}

// FOLLOW: fused launch — the wave's chains are all of one channel (q = channel * numPackets + packet), and rows are
// only read once the entropy lane of their packet has published them.
// PPW != 64 (k_dec_fused_wg): `block` = workgroup * 3 + r, the r-th of the three predictor waves behind the workgroup's
// entropy wave; its chains are slots r * 32 .. r * 32 + 31 of the workgroup's PPW first-channel chains followed by its PPW
// second-channel chains, and the progress words are the workgroup's (LDS).
template <bool FOLLOW, int PPW = 64>
__device__ __forceinline__ void unpc_fast_body(const DecV1Args &V, uint32_t block, const uint32_t *progLds = nullptr)
{
    constexpr bool LOCAL = PPW != 64;
    const DecodeArgs &A = V.d;
    const int lane = threadIdx.x & 63;
    const uint64_t q = (uint64_t)block * 32u + lane / 2;
    bool inRange = q < (uint64_t)A.numPackets * A.numChannels;
    uint32_t p, ch, slotLocal = 0;
    if constexpr (LOCAL) {
        const uint32_t wg = block / 3u, s = (block % 3u) * 32u + (uint32_t)lane / 2;
        ch = s / (uint32_t)PPW;
        slotLocal = s % (uint32_t)PPW;
        p = wg * (uint32_t)PPW + slotLocal;
        inRange = ch < A.numChannels && p < A.numPackets;
        if (!inRange) p = ch = 0;
    } else if (FOLLOW) {
        ch = inRange ? (uint32_t)(q / A.numPackets) : 0;
        p = inRange ? (uint32_t)(q % A.numPackets) : 0;
    } else {
        p = inRange ? (uint32_t)(q / A.numChannels) : 0;
        ch = inRange ? (uint32_t)(q % A.numChannels) : 0;
    }
    const uint64_t chain = (uint64_t)p * A.numChannels + ch;
    const DecRec *rec = A.recs + p;
    const bool active = inRange && unpc_fast_ok(A, rec, ch);
    if (!__any(active)) return;
    const uint32_t n = active ? rec->numSamples : 0;
    // rows < `rows` (capped at the lane's own length) must have been published before they are loaded
    const uint32_t *progPtr = LOCAL ? progLds + slotLocal * 2 + ch : V.prog + (uint64_t)p * 2 + ch;
    uint32_t availMin = 0;
    auto need = [&](uint32_t rows) {
        if constexpr (FOLLOW) {
            if (availMin >= rows) return;
            const uint32_t want = active ? min(rows, n) : 0u;
            bool seen = false;
            uint32_t a = 0;
            for (uint32_t spins = 0; spins < A.ho.spinLimit; spins++) {
                if constexpr (LOCAL)
                    a = active ? __hip_atomic_load(progPtr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0xffffffffu;
                else
                    a = active ? __hip_atomic_load(progPtr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                if (__all(a >= want)) {
                    availMin = wave_min_dec(a);
                    seen = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            if (!seen) {
                // the entropy lanes never published these rows: what is in the plane is not this packet's residuals.
                // The packets concerned fail (kALAC_ParamError), the context's error word makes the call fail, and the
                // wave stops waiting so that the launch drains.
                if (active && a < want) A.statusOut[p] = -50;
                if (A.ho.err && lane == 0) __hip_atomic_store(A.ho.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                availMin = 0xffffffffu;
            }
            if constexpr (LOCAL)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            else
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    };
    need(16);
    const int na = active ? (int)rec->c[ch].num : 4;
    const uint32_t chanbits = A.bitDepth - (active ? rec->bytesShifted : 0) * 8 + ((active ? rec->elementChannels : 1) == 2 ? 1 : 0);
    int32_t *row = V.plane + (inRange ? chain : 0) * A.frameSize;
    const bool writer = active && (lane & 1) == 0;

    // ---- head: both lanes of the pair compute the same 16 samples, no exchange needed to set up the windows ----
    int32_t d16[16], o16[16], a8[8];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int4 t = active ? ((const int4 *)row)[q] : make_int4(0, 0, 0, 0);
        d16[4 * q] = t.x;
        d16[4 * q + 1] = t.y;
        d16[4 * q + 2] = t.z;
        d16[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) a8[k] = (active && k < na) ? (int32_t)rec->c[ch].coefs[k] : 0;
    if (na == 8) unpc_head16<8>(d16, o16, a8, 32 - chanbits);
    else unpc_head16<4>(d16, o16, a8, 32 - chanbits);
    if (writer) {
#pragma unroll
        for (int q = 0; q < 4; q++) ((int4 *)row)[q] = make_int4(o16[4 * q], o16[4 * q + 1], o16[4 * q + 2], o16[4 * q + 3]);
    }
    // (measured and not kept: hiding the lane-constant AND masks of the step from the compiler turns its v_mov_b32_dpp +
    // select pairs into v_and_b32_dpp — 59 -> 57 instructions per step — and the launch gets SLOWER, 1.655 -> 1.731 ms at
    // 10 000 packets: the fused form puts the DPP wait states on the dependent chain)
    const LmsDecLane L = make_dec_lane(lane, na);
    int32_t a[4], w[4], tp;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = L.h ? a8[4 + i] : a8[i];
        w[i] = (L.h ? o16[11 - i] : o16[15 - i]) & L.feed;
    }
    tp = (na == 8 ? o16[7] : o16[11]) & L.feed;

    // ---- 8-step blocks; the residuals of block i+1 are loaded while block i computes ----
    const uint32_t nMax = wave_max_u32(n);
    auto load8 = [&](uint32_t jb, int32_t (&d)[8]) {
        // rows are frameSize long and the plane is padded, so the (unused) over-read of the last block stays inside
        const int4 t0 = *(const int4 *)(row + jb), t1 = *(const int4 *)(row + jb + 4);
        d[0] = t0.x; d[1] = t0.y; d[2] = t0.z; d[3] = t0.w;
        d[4] = t1.x; d[5] = t1.y; d[6] = t1.z; d[7] = t1.w;
    };
    int32_t dA[8], dB[8];
    need(min(24u, nMax));
    load8(16, dA);
    auto step8 = [&](uint32_t jb, const int32_t (&cur)[8], int32_t (&nxt)[8]) {
        need(min(jb + 16, nMax));
        load8(jb + 8, nxt);
        int32_t o[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++) o[s2] = lms4_step_dec(a, w, tp, cur[s2], L, chanbits);
        if (writer && jb < n) {
            *(int4 *)(row + jb) = make_int4(o[0], o[1], o[2], o[3]);
            *(int4 *)(row + jb + 4) = make_int4(o[4], o[5], o[6], o[7]);
        }
    };
    for (uint32_t jb = 16; jb < nMax; jb += 16) {
        step8(jb, dA, dB);
        if (jb + 8 < nMax) step8(jb + 8, dB, dA);
    }
}


// ---- separate launches (large batches): ONE lane per chain, chains sorted by tap count -------------------------------
// Where every kernel fills the machine by itself the cost of the predictor is wave-instructions per chain step, and
// the two-lane mapping above spends 58 per 32 chains.  Here a lane holds all T taps of its chain (T = 4 or 8 = the
// chain's own tap count, so there are no dead taps, no cross-lane exchange and the weights are compile-time constants)
// and a wave walks 64 chains.  k_dec_header has sorted the chains the fast path accepts by tap count into ONE list filled
// from both ends (dec_lists), and the packets whose two chains both qualify into the pair list.
constexpr uint32_t kDecDirectPackets = 80000;  // separate launches read the caller's stream directly from here on (decode_v1_pass)
// One fused launch up to here, separate launches above: 32 768 stereo packets (30 000: 2.78 against 2.97 ms, 34 000: 3.28 against
// 3.33 at 24 bits) — but a mono packet is half the entropy work behind the same workgroup count, and mono streams cross over
// where the fused workgroups (one per 48 packets) reach four per CU: 45 000 packets 1.98 against 2.00 ms, 50 000 2.30 against
// 2.07 (profiles/r04/decode_direct_sweep.log).  Until the end of round 4 mono used the stereo rule (65 536 chains).
constexpr uint64_t kDecFusedChains = 65536, kDecFusedChainsMono = 49152;
__host__ inline bool dec_fused_auto(uint32_t numPackets, uint32_t numChannels)
{
    return (uint64_t)numPackets * numChannels <= (numChannels == 1 ? kDecFusedChainsMono : kDecFusedChains);
}

// one unpc step of a lane that holds all T taps: returns out[j]; updates a[], the window w[] and tp
template <int T>
__device__ __forceinline__ int32_t lms_step_dec_wide(int32_t (&a)[T], int32_t (&w)[T], int32_t &tp, int32_t del, uint32_t chanbits)
{
    int32_t b[T];
#pragma unroll
    for (int i = 0; i < T; i++) b[i] = tp - w[i];
    int32_t s = 255;  // -((256 - s) >> 9) == (s + 255) >> 9
#pragma unroll
    for (int i = 0; i < T; i++) s = __mul24((int32_t)(int16_t)a[i], b[i]) + s;
    const int32_t out = __builtin_amdgcn_sbfe(del + tp - (s >> kDenShift), 0, chanbits);
    // coefficient update, driven by del: the threshold form of the walk (alac_lms.hpp), weights T - i
    const int32_t nd = -del;
    const int32_t adel = max(del, nd);
    const int32_t nsg = sign3(nd);
    const int32_t rc = (del >> 31) & ((1 << kDenShift) - 1);
    int32_t sb[T];
    uint32_t t[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
        sb[i] = sign3(b[i]);
        t[i] = (uint32_t)(__mul24(sb[i], b[i]) + rc) >> kDenShift;
    }
    int32_t S[T];
    S[T - 1] = 0;
#pragma unroll
    for (int i = T - 1; i > 0; i--) S[i - 1] = (int32_t)__umul24(t[i], (uint32_t)(T - i)) + S[i];
#pragma unroll
    for (int i = 0; i < T; i++) a[i] = __mul24(adel > S[i] ? nsg : 0, sb[i]) + a[i];
    tp = w[T - 1];
#pragma unroll
    for (int i = T - 1; i > 0; i--) w[i] = w[i - 1];
    w[0] = out;
    return out;
}

// ---- residual blocks of the one-lane-per-chain predictors ----
// 32-step blocks = one 128-byte line of the lane's row: the eight 16-byte loads of a block are issued back to back, so the
// line is fetched from L2 once (with two loads per 8 steps the line had left the L1 — 10 waves x 64 rows — before its next
// quarter was wanted: four L2 requests per line).  The residuals of block i + 1 are loaded while block i computes.
// (Round 4 also built a COOPERATIVE form — instruction q fetches, for lane L, 16 bytes of the row of lane 8 q + L / 8, so that
// eight neighbouring lanes cover one whole line, and the block is transposed through LDS — because tools/fetch_calibrate.hip
// shows a lane streaming through its own row with ONE 16-byte load per iteration fetching every line 2.25 times.  It changed
// nothing: 5.13 -> 5.14 ms at 125 000 packets and FETCH_SIZE of k_dec_unpc_wide 1.839 -> 1.835 GB for 3.6 GB of distinct rows —
// with the eight loads back to back the lines already arrive as whole 128-byte requests, once each, and the kernel issues at
// the chip's instruction ceiling.  Removed.)
__device__ __forceinline__ void load32(const int32_t *row, uint32_t jb, int32_t (&d)[32])
{
    // rows are frameSize long and the plane is padded by 256 bytes: the (unused) over-read of the last block, at most
    // 31 samples past the longest row, stays inside
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int4 t = *(const int4 *)(row + jb + 4 * q);
        d[4 * q] = t.x;
        d[4 * q + 1] = t.y;
        d[4 * q + 2] = t.z;
        d[4 * q + 3] = t.w;
    }
}

template <int T>
__device__ __forceinline__ void unpc_wide_body(const DecV1Args &V, uint32_t block, uint32_t count)
{
    const DecodeArgs &A = V.d;
    const uint32_t total = A.numPackets * A.numChannels;
    if (block * 64u >= count) return;
    const uint32_t idx = block * 64u + (threadIdx.x & 63);
    const bool active = idx < count;
    const uint32_t *chains = dec_lists(V).chains;
    const uint32_t chain = active ? (T == 8 ? chains[total - 1 - idx] : chains[idx]) : 0;
    const uint32_t p = chain / A.numChannels, ch = chain % A.numChannels;
    const DecRec *rec = A.recs + p;
    const uint32_t n = active ? rec->numSamples : 0;
    const uint32_t chanbits = A.bitDepth - (active ? rec->bytesShifted : 0) * 8 + ((active ? rec->elementChannels : 1) == 2 ? 1 : 0);
    int32_t *row = V.plane + (uint64_t)chain * A.frameSize;

    // ---- head: the first 16 samples lane-serially (warm-up positions + the first regular steps) ----
    int32_t d16[16], o16[16], a8[8];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int4 t = active ? ((const int4 *)row)[q] : make_int4(0, 0, 0, 0);
        d16[4 * q] = t.x;
        d16[4 * q + 1] = t.y;
        d16[4 * q + 2] = t.z;
        d16[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) a8[k] = (active && k < T) ? (int32_t)rec->c[ch].coefs[k] : 0;
    unpc_head16<T>(d16, o16, a8, 32 - chanbits);
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; q++) ((int4 *)row)[q] = make_int4(o16[4 * q], o16[4 * q + 1], o16[4 * q + 2], o16[4 * q + 3]);
    }
    int32_t a[T], w[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
        a[i] = a8[i];
        w[i] = o16[15 - i];  // out[j - 1 - i] at j = 16
    }
    int32_t tp = o16[15 - T];  // out[j - T - 1]

    // ---- 32-step blocks (load32) ----
    const uint32_t nMax = wave_max_u32(n);
    int32_t dA[32], dB[32];
    if (32 < nMax) load32(row, 32, dA);
    if (16 < nMax) {
        // samples 16 .. 31 on their own, so that the 32-step blocks start on a line boundary (rows of the usual frame sizes
        // are 128-byte aligned)
        int32_t d[16], o[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int4 t = *(const int4 *)(row + 16 + 4 * q);
            d[4 * q] = t.x;
            d[4 * q + 1] = t.y;
            d[4 * q + 2] = t.z;
            d[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) o[s2] = lms_step_dec_wide<T>(a, w, tp, d[s2], chanbits);
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (active && 16 + 4 * q < n) *(int4 *)(row + 16 + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
    auto step32 = [&](uint32_t jb, const int32_t (&cur)[32], int32_t (&nxt)[32]) {
        if (jb + 32 < nMax) load32(row, jb + 32, nxt);
        int32_t o[32];
#pragma unroll
        for (int s2 = 0; s2 < 32; s2++) o[s2] = lms_step_dec_wide<T>(a, w, tp, cur[s2], chanbits);
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (active && jb + 4 * q < n) *(int4 *)(row + jb + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    };
    for (uint32_t jb = 32; jb < nMax; jb += 64) {
        step32(jb, dA, dB);
        if (jb + 32 < nMax) step32(jb + 32, dB, dA);
    }
}

// ---- any tap count 1..8, any denShift 1..15, mode 0 (round 4): the chains of OTHER encoders' streams on the one-lane mapping.
// An 8-tap lane with the lane's own count na: taps i >= na are dead (coefficient 0, sign forced to 0: t_i = 0, nothing is ever
// added), the "top" sample is w[na - 1] as it leaves the window (an 8-way select: the compiler keeps the eight lane masks in
// scalar registers), the threshold weights are na - i, the shifts and the rounding are the lane's denShift.  No term of the
// walk can wrap: chanBits <= 23 (k_dec_header), so |b| < 2^23 and sum_i (na - i) ((|b_i| + rc) >> ds) < 2^29.
struct AnyLane {
    int32_t na, ds, denhalf, rcMask;
    int32_t am[8];    // -1 for live taps
    uint32_t wg[8];   // na - i (0 for dead taps)
};
__device__ __forceinline__ int32_t lms_step_dec_any(int32_t (&a)[8], int32_t (&w)[8], int32_t &tp, int32_t del, uint32_t chanbits,
                                                    const AnyLane &L)
{
    int32_t b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = tp - w[i];
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s = __mul24((int32_t)(int16_t)a[i], b[i]) + s;  // (dead taps: a = 0)
    const int32_t out = __builtin_amdgcn_sbfe(del + tp + ((L.denhalf - s) >> L.ds), 0, chanbits);
    const int32_t nd = -del;
    const int32_t adel = max(del, nd);
    const int32_t nsg = sign3(nd);
    const int32_t rc = (del >> 31) & L.rcMask;
    int32_t sb[8];
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        sb[i] = sign3(b[i]) & L.am[i];
        t[i] = (uint32_t)(__mul24(sb[i], b[i]) + rc) >> L.ds;
    }
    int32_t S[8];
    S[7] = 0;
#pragma unroll
    for (int i = 7; i > 0; i--) S[i - 1] = (int32_t)__umul24(t[i], L.wg[i]) + S[i];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = __mul24(adel > S[i] ? nsg : 0, sb[i]) + a[i];
    int32_t leaving = w[0];
#pragma unroll
    for (int i = 1; i < 8; i++) leaving = (L.na - 1 == i) ? w[i] : leaving;
    tp = leaving;
#pragma unroll
    for (int i = 7; i > 0; i--) w[i] = w[i - 1];
    w[0] = out;
    return out;
}

__device__ __forceinline__ void unpc_any_body(const DecV1Args &V, uint32_t block, uint32_t count)
{
    const DecodeArgs &A = V.d;
    if (block * 64u >= count) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t idx = block * 64u + lane;
    const bool active = idx < count;
    const uint32_t chain = active ? dec_lists(V).any[idx] : 0;
    const uint32_t p = chain / A.numChannels, ch = chain % A.numChannels;
    const DecRec *rec = A.recs + p;
    const uint32_t n = active ? rec->numSamples : 0;
    const uint32_t chanbits = A.bitDepth - (active ? rec->bytesShifted : 0) * 8 + ((active ? rec->elementChannels : 1) == 2 ? 1 : 0);
    int32_t *row = V.plane + (uint64_t)chain * A.frameSize;
    AnyLane L;
    L.na = active ? (int32_t)rec->c[ch].num : 1;
    L.ds = active ? (int32_t)rec->c[ch].denShift : 9;
    L.denhalf = 1 << (L.ds - 1);
    L.rcMask = (1 << L.ds) - 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        L.am[i] = i < L.na ? -1 : 0;
        L.wg[i] = i < L.na ? (uint32_t)(L.na - i) : 0u;
    }
    int32_t a[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = (active && i < L.na) ? (int32_t)rec->c[ch].coefs[i] : 0;
        w[i] = 0;
    }
    int32_t tp = 0;
    // ---- the first 16 samples: position 0 as it comes, positions 1 .. na first-order (codec/dp_dec.c:97-101), then regular steps;
    // both forms are evaluated and selected per lane (16 steps, once per chain)
    {
        int32_t d[16], o[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int4 t = active ? ((const int4 *)row)[q] : make_int4(0, 0, 0, 0);
            d[4 * q] = t.x;
            d[4 * q + 1] = t.y;
            d[4 * q + 2] = t.z;
            d[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j == 0) {
                o[0] = d[0];
                w[0] = o[0];
            } else {
                // warm-up form: the window shifts, nothing adapts; regular form: the full step.  Both from the same state.
                int32_t a2[8], w2[8], tp2 = tp;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    a2[i] = a[i];
                    w2[i] = w[i];
                }
                const int32_t reg = lms_step_dec_any(a2, w2, tp2, d[j], chanbits, L);
                const int32_t warm = __builtin_amdgcn_sbfe(d[j] + w[0], 0, chanbits);
                const bool isWarm = j <= L.na;
                int32_t leaving = w[0];
#pragma unroll
                for (int i = 1; i < 8; i++) leaving = (L.na - 1 == i) ? w[i] : leaving;
                o[j] = isWarm ? warm : reg;
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = isWarm ? a[i] : a2[i];
#pragma unroll
                for (int i = 7; i > 0; i--) w[i] = w[i - 1];
                w[0] = o[j];
                tp = leaving;  // (the regular step leaves the same sample: w_old[na - 1])
            }
        }
        if (active) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (4u * q < n) ((int4 *)row)[q] = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
    }
    const uint32_t nMax = wave_max_u32(n);
    int32_t dA[32], dB[32];
    if (32 < nMax) load32(row, 32, dA);
    if (16 < nMax) {
        int32_t d[16], o[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int4 t = *(const int4 *)(row + 16 + 4 * q);
            d[4 * q] = t.x;
            d[4 * q + 1] = t.y;
            d[4 * q + 2] = t.z;
            d[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) o[s2] = lms_step_dec_any(a, w, tp, d[s2], chanbits, L);
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (active && 16 + 4 * q < n) *(int4 *)(row + 16 + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
    auto step32 = [&](uint32_t jb, const int32_t (&cur)[32], int32_t (&nxt)[32]) {
        if (jb + 32 < nMax) load32(row, jb + 32, nxt);
        int32_t o[32];
#pragma unroll
        for (int s2 = 0; s2 < 32; s2++) o[s2] = lms_step_dec_any(a, w, tp, cur[s2], chanbits, L);
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (active && jb + 4 * q < n) *(int4 *)(row + jb + 4 * q) = make_int4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    };
    for (uint32_t jb = 32; jb < nMax; jb += 64) {
        step32(jb, dA, dB);
        if (jb + 32 < nMax) step32(jb + 32, dB, dA);
    }
}

// ---- pair mode: the two chains of a stereo packet in adjacent lanes; the lanes un-mix and write the PCM themselves ----
// T = 4: both chains have 4 taps.  T = 8: at least one has 8; a 4-tap chain then lives in an 8-tap lane with dead upper
// taps: their coefficients start at 0 and stay there (the sign of b is forced to 0, so t = 0 and nothing is added), the
// "top" sample is the one five back instead of nine, and the threshold weights are na - i with the lane's own na.
template <int T>
__device__ __forceinline__ int32_t lms_step_dec_pair(int32_t (&a)[T], int32_t (&w)[T], int32_t &tp, int32_t del, uint32_t chanbits,
                                                     bool is4, int32_t act, const uint32_t (&wg)[4])
{
    int32_t b[T];
#pragma unroll
    for (int i = 0; i < T; i++) b[i] = tp - w[i];
    int32_t s = 255;
#pragma unroll
    for (int i = 0; i < T; i++) s = __mul24((int32_t)(int16_t)a[i], b[i]) + s;
    const int32_t out = __builtin_amdgcn_sbfe(del + tp - (s >> kDenShift), 0, chanbits);
    const int32_t nd = -del;
    const int32_t adel = max(del, nd);
    const int32_t nsg = sign3(nd);
    const int32_t rc = (del >> 31) & ((1 << kDenShift) - 1);
    int32_t sb[T];
    uint32_t t[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
        sb[i] = sign3(b[i]);
        if (T == 8 && i >= 4) sb[i] &= act;
        t[i] = (uint32_t)(__mul24(sb[i], b[i]) + rc) >> kDenShift;
    }
    int32_t S[T];
    S[T - 1] = 0;
#pragma unroll
    for (int i = T - 1; i > 0; i--) {
        const uint32_t weight = T == 4 ? (uint32_t)(T - i) : (i < 4 ? wg[i] : (uint32_t)(T - i));
        S[i - 1] = (int32_t)__umul24(t[i], weight) + S[i];
    }
#pragma unroll
    for (int i = 0; i < T; i++) a[i] = __mul24(adel > S[i] ? nsg : 0, sb[i]) + a[i];
    tp = T == 4 ? w[3] : (is4 ? w[3] : w[7]);
#pragma unroll
    for (int i = T - 1; i > 0; i--) w[i] = w[i - 1];
    w[0] = out;
    return out;
}

template <int T, int DEPTH = 16>
__device__ __forceinline__ void unpc_pair_body(const DecV1Args &V, uint32_t block, uint32_t count)
{
    const DecodeArgs &A = V.d;
    if (block * 32u >= count) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t idx = block * 32u + lane / 2, ch = lane & 1;
    const uint32_t *pairs = dec_lists(V).pairs;
    const uint32_t p = idx < count ? (T == 8 ? pairs[A.numPackets - 1 - idx] : pairs[idx]) : 0;
    const DecRec *rec = A.recs + p;
    const bool active = idx < count && rec->status == 0;  // a packet the entropy lane gave up on keeps its PCM untouched
    const uint32_t n = active ? rec->numSamples : 0;
    const uint32_t chanbits = A.bitDepth - (active ? rec->bytesShifted : 0) * 8 + 1;
    const bool is4 = T == 4 || !active || rec->c[ch].num == 4;
    const int32_t act = is4 ? 0 : -1;
    uint32_t wg[4];
#pragma unroll
    for (int i = 0; i < 4; i++) wg[i] = (is4 ? 4u : 8u) - (uint32_t)i;
    const int32_t mixRes = active ? rec->mixRes : 0, mixBits = active ? rec->mixBits : 0;
    const int32_t mixMask = mixRes != 0 ? -1 : 0;
    const bool isU = ch == 0;
    int32_t *row = V.plane + ((uint64_t)p * 2 + ch) * A.frameSize;
    uint32_t *pcm = (uint32_t *)(A.pcmOut + (uint64_t)p * A.frameSize * 4);  // one word per frame: L | R << 16
    // 20 / 24 bits: six bytes per frame; the shifted-off bytes (one per sample: k_dec_header lists no other packet as a pair)
    // sit in the staged stream, two per frame from bit shiftPos on (codec/ALACDecoder.cu:905-915, gpu_unmix24 :282-338)
    uint8_t *pcm3 = A.pcmOut + (uint64_t)p * A.frameSize * 6;
    const bool shifted = DEPTH == 24 && active && rec->bytesShifted != 0;
    const uint64_t pktOff = active ? A.offsets[p] : 0;
    const uint64_t sWordBase = pktOff >> 2, sBit0 = (pktOff & 3) * 8 + (active ? rec->shiftPos : 0);
    const uint64_t sTail = dec_tail_start(V);

    // K outputs of each lane of a pair -> K frames: the U lane takes the first K / 2 frames, the V lane the rest; one DPP
    // exchange per frame hands each lane the sample of the other channel it needs (gpu_unmix16, codec/ALACDecoder.cu:193-223)
    typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
    auto emit = [&](auto &o, auto kc, uint32_t jb) {
        constexpr int K = decltype(kc)::value;
        if constexpr (DEPTH != 16) {
            // this lane's K / 2 frames: their shift bytes are K bytes of the stream = K / 4 words at an arbitrary bit offset
            const uint32_t f0 = jb + (isU ? 0u : (uint32_t)(K / 2));
            uint32_t sw[K / 4 + 1];
            uint32_t ssh = 0;
            if (shifted) {
                const uint64_t b = sBit0 + (uint64_t)f0 * 16;
                const uint64_t i0 = min(sWordBase + (b >> 5), V.capWords - (uint64_t)(K / 4 + 2));  // status-0 packets lie inside the stage
                ssh = (uint32_t)(b & 31);
#pragma unroll
                for (int q = 0; q <= K / 4; q++) sw[q] = dec_word(V, sTail, i0 + q);
            }
            uint32_t fld[K];  // L0 R0 L1 R1 ... as 24-bit little-endian fields
#pragma unroll
            for (int k = 0; k < K / 2; k++) {
                const int32_t give = isU ? o[k + K / 2] : o[k];
                const int32_t recv = dpp_xor1(give);
                const int32_t mine = isU ? o[k] : o[k + K / 2];
                const int32_t uu = isU ? mine : recv, vv = isU ? recv : mine;
                int32_t l = uu + ((vv - ((mixRes * vv) >> mixBits)) & mixMask);
                int32_t r = mixRes != 0 ? l - vv : vv;
                if constexpr (DEPTH == 24) {
                    if (shifted) {
                        // frame k's two bytes = bits [16 k, 16 k + 16) of the funnel-shifted words
                        const uint32_t wq = (uint32_t)(k >> 1);
                        const uint32_t x32 = ssh ? (sw[wq] << ssh) | (sw[wq + 1] >> (32 - ssh)) : sw[wq];
                        const uint32_t xx = (k & 1) ? (x32 & 0xffffu) : (x32 >> 16);
                        l = (int32_t)(((uint32_t)l << 8) | (xx >> 8));
                        r = (int32_t)(((uint32_t)r << 8) | (xx & 0xffu));
                    }
                } else {
                    l = (int32_t)((uint32_t)l << 4);  // 20 bits, left-justified in three bytes (gpu_unmix20 :225-280)
                    r = (int32_t)((uint32_t)r << 4);
                }
                fld[2 * k] = (uint32_t)l & 0xffffffu;
                fld[2 * k + 1] = (uint32_t)r & 0xffffffu;
            }
            uint8_t *dst = pcm3 + (uint64_t)f0 * 6;
            if (active && f0 + (uint32_t)(K / 2) <= n) {
                // four 3-byte fields make three words; K fields = 3 K / 4 words = 3 K / 16 sixteen-byte stores
                typedef uint32_t U4s __attribute__((ext_vector_type(4), aligned(4)));
                uint32_t d[3 * K / 4];
#pragma unroll
                for (int g = 0; g < K / 4; g++) {
                    const uint32_t a0 = fld[4 * g], a1 = fld[4 * g + 1], a2 = fld[4 * g + 2], a3 = fld[4 * g + 3];
                    d[3 * g] = a0 | (a1 << 24);
                    d[3 * g + 1] = (a1 >> 8) | (a2 << 16);
                    d[3 * g + 2] = (a2 >> 16) | (a3 << 8);
                }
#pragma unroll
                for (int q = 0; q < 3 * K / 16; q++) {
                    const U4s t4 = {d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
                    *(U4s *)(dst + 16 * q) = t4;
                }
            } else if (active) {
                // the last frames of a short packet
#pragma unroll
                for (int k = 0; k < K / 2; k++) {
                    if (f0 + (uint32_t)k < n) {
#pragma unroll
                        for (int c2 = 0; c2 < 2; c2++) {
                            const uint32_t v3 = fld[2 * k + c2];
                            dst[6 * k + 3 * c2] = (uint8_t)v3;
                            dst[6 * k + 3 * c2 + 1] = (uint8_t)(v3 >> 8);
                            dst[6 * k + 3 * c2 + 2] = (uint8_t)(v3 >> 16);
                        }
                    }
                }
            }
            return;
        }
        uint32_t word[K / 2];
#pragma unroll
        for (int k = 0; k < K / 2; k++) {
            const int32_t give = isU ? o[k + K / 2] : o[k];
            const int32_t recv = dpp_xor1(give);
            const int32_t mine = isU ? o[k] : o[k + K / 2];
            const int32_t uu = isU ? mine : recv, vv = isU ? recv : mine;
            const int32_t l = uu + ((vv - ((mixRes * vv) >> mixBits)) & mixMask);
            const int32_t r = mixRes != 0 ? l - vv : vv;
            word[k] = __builtin_amdgcn_perm((uint32_t)r, (uint32_t)l, 0x05040100u);
        }
        const uint32_t f0 = jb + (isU ? 0u : (uint32_t)(K / 2));
#pragma unroll
        for (int q = 0; q < K / 8; q++) {
            const uint32_t f = f0 + 4 * q;
            if (active && f + 4 <= n) {
                const U4 t4 = {word[4 * q], word[4 * q + 1], word[4 * q + 2], word[4 * q + 3]};
                *(U4 *)(pcm + f) = t4;
            } else if (active && f < n) {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (f + e < n) pcm[f + e] = word[4 * q + e];
            }
        }
    };

    // ---- head: the first 16 samples lane-serially, with the lane's own tap count ----
    int32_t d16[16], o16[16], a8[8];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int4 t = active ? ((const int4 *)row)[q] : make_int4(0, 0, 0, 0);
        d16[4 * q] = t.x;
        d16[4 * q + 1] = t.y;
        d16[4 * q + 2] = t.z;
        d16[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) a8[k] = (active && k < (is4 ? 4 : 8)) ? (int32_t)rec->c[ch].coefs[k] : 0;
    if constexpr (T == 4) {
        unpc_head16<4>(d16, o16, a8, 32 - chanbits);
    } else {
        int32_t o4[16], c4[8], o8[16], c8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c4[k] = c8[k] = a8[k];
        unpc_head16<4>(d16, o4, c4, 32 - chanbits);
        unpc_head16<8>(d16, o8, c8, 32 - chanbits);
#pragma unroll
        for (int k = 0; k < 16; k++) o16[k] = is4 ? o4[k] : o8[k];
#pragma unroll
        for (int k = 0; k < 8; k++) a8[k] = is4 ? (k < 4 ? c4[k] : 0) : c8[k];
    }
    emit(o16, std::integral_constant<int, 16>{}, 0);
    int32_t a[T], w[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
        a[i] = a8[i];
        w[i] = o16[15 - i];
    }
    int32_t tp = T == 4 ? o16[11] : (is4 ? o16[11] : o16[7]);

    const uint32_t nMax = wave_max_u32(n);
    int32_t dA[32], dB[32];
    if (32 < nMax) load32(row, 32, dA);
    if (16 < nMax) {
        int32_t d[16], o[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int4 t = *(const int4 *)(row + 16 + 4 * q);
            d[4 * q] = t.x;
            d[4 * q + 1] = t.y;
            d[4 * q + 2] = t.z;
            d[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) o[s2] = lms_step_dec_pair<T>(a, w, tp, d[s2], chanbits, is4, act, wg);
        emit(o, std::integral_constant<int, 16>{}, 16);
    }
    auto step32 = [&](uint32_t jb, const int32_t (&cur)[32], int32_t (&nxt)[32]) {
        if (jb + 32 < nMax) load32(row, jb + 32, nxt);
        int32_t o[32];
#pragma unroll
        for (int s2 = 0; s2 < 32; s2++) o[s2] = lms_step_dec_pair<T>(a, w, tp, cur[s2], chanbits, is4, act, wg);
        emit(o, std::integral_constant<int, 32>{}, jb);
    };
    for (uint32_t jb = 32; jb < nMax; jb += 64) {
        step32(jb, dA, dB);
        if (jb + 32 < nMax) step32(jb + 32, dB, dA);
    }
}

// ONE launch for both lists: the 8-tap waves (the longer chains) take the first workgroups, the 4-tap waves the ones behind.
// As two launches, one after the other, each ran at about two waves per SIMD (2 266 and 1 640 waves at 125 000 stereo
// packets) and was as slow as its own serial chain: 1.65 + 1.29 ms.
// (pair mode: class-B pairs, 8-tap chains, class-A pairs, 4-tap chains)
// Four waves to a workgroup (one per SIMD of its CU), consecutive roles: see k_dec_entropy_wide.
// DEPTH: what the pairs write (16: one word per frame; 20 / 24: six bytes per frame) — one instantiation per output format,
// so that a launch carries four loop bodies, not eight (they share the CU's instruction cache)
template <int DEPTH>
__global__ __launch_bounds__(64 * kEntWavesPerWg) void k_dec_unpc_wide(DecV1Args V)
{
    const uint32_t *cnt = dec_lists(V).cnt;
    const uint32_t c4 = cnt[0], c8 = cnt[1], pA = cnt[2], pB = cnt[3], cAny = cnt[6];
    const uint32_t nbB = (pB + 31u) / 32u, nb8 = (c8 + 63u) / 64u, nbA = (pA + 31u) / 32u, nb4 = (c4 + 63u) / 64u;
    uint32_t b = blockIdx.x * (uint32_t)kEntWavesPerWg + (threadIdx.x >> 6);
    if (b < nbB) return unpc_pair_body<8, DEPTH>(V, b, pB);
    b -= nbB;
    if (b < nb8) return unpc_wide_body<8>(V, b, c8);
    b -= nb8;
    if (b < nbA) return unpc_pair_body<4, DEPTH>(V, b, pA);
    b -= nbA;
    if (b < nb4) return unpc_wide_body<4>(V, b, c4);
    b -= nb4;
    unpc_any_body(V, b, cAny);  // other encoders' chains (none in a stream of this library's or Apple's encoder)
}

// ---- fused launch: the followers of an entropy wave are waves of ITS workgroup.  A workgroup = the entropy
// wave of kFusedPpw = 48 packets + the three predictor waves of their 96 chains (first-channel chains, then second-channel
// chains, 32 to a wave), one wave per SIMD of one CU.  Producer and followers share that CU's L1 and L2, so a publish is a
// workgroup-scope release (wait for the stores to leave the wave) and two LDS words per packet.  (Rounds 2-3 had the
// followers anywhere on the chip behind progress words in HBM: every publish was an agent-scope release — an L2 write-back of
// ~10 us, sixteen times per packet; removed in round 4.)  Workgroups >= nEnt: one wave per packet for the uncompressed
// elements (nobody waits for them; their dispatch hides under the entropy chain instead of costing a launch of its own).
constexpr int kFusedPpw = 48;
__global__ __launch_bounds__(256, 1) void k_dec_fused_wg(DecV1Args V, uint32_t nEnt)
{
    __shared__ uint32_t ringOne[64 * kWinStride];
    __shared__ uint32_t progLds[kFusedPpw * 2];
    const uint32_t slot = threadIdx.x >> 6;
    if (blockIdx.x < nEnt) {
        if (threadIdx.x < kFusedPpw * 2) progLds[threadIdx.x] = 0;
        __syncthreads();
        if (slot == 0)
            entropy_body<true, false, false, kFusedPpw>(V, ringOne, blockIdx.x, progLds);
        else
            unpc_fast_body<true, kFusedPpw>(V, blockIdx.x * 3u + (slot - 1), progLds);
    } else {
        const uint32_t p = (blockIdx.x - nEnt) * 4u + slot;
        if (p < V.d.numPackets) raw_body<false>(V, p, threadIdx.x & 63, 64);
    }
}

__global__ __launch_bounds__(64) void k_dec_unpc(DecV1Args V)
{
    const DecodeArgs &A = V.d;
    const uint64_t gid = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (gid >= (uint64_t)A.numPackets * A.numChannels) return;
    const uint32_t p = (uint32_t)(gid / A.numChannels), ch = (uint32_t)(gid % A.numChannels);
    const DecRec *rec = A.recs + p;
    if (rec->status != 0 || rec->escape || rec->elementChannels == 0) return;
    if (unpc_fast_ok(A, rec, ch)) return;  // the fast predictor bodies'
    if (V.lists && unpc_any_ok(A, rec, ch)) return;  // unpc_any_body's (separate launches)
    const uint32_t chanbits = A.bitDepth - rec->bytesShifted * 8 + (rec->elementChannels == 2 ? 1 : 0);
    int32_t *row = V.plane + ((uint64_t)p * A.numChannels + ch) * A.frameSize;
    const DecChan &c = rec->c[ch];
    // :829-838: mode != 0 runs the first-order pass first
    if (c.mode != 0) unpc_first_order(row, 1, rec->numSamples, 32 - chanbits);
    unpc_any(row, 1, rec->numSamples, c.coefs, c.num, chanbits, c.denShift);
}

// ---- un-mix + pack (gpu_unmixNN / gpu_copyPredictorToNN, codec/ALACDecoder.cu:193-495) ----
template <int DEPTH>
__device__ __forceinline__ void put_sample(uint8_t *q, int32_t x)
{
    if constexpr (DEPTH == 16) {
        *(int16_t *)q = (int16_t)x;
    } else if constexpr (DEPTH == 32) {
        *(int32_t *)q = x;
    } else {
        if constexpr (DEPTH == 20) x = (int32_t)((uint32_t)x << 4);
        q[0] = (uint8_t)x;
        q[1] = (uint8_t)(x >> 8);
        q[2] = (uint8_t)(x >> 16);
    }
}

template <int DEPTH, int CH>
__device__ __forceinline__ void unmix_part(const DecV1Args &V, uint32_t p, uint32_t part, uint32_t bx)
{
    const DecodeArgs &A = V.d;
    const DecRec *rec = A.recs + p;
    if (rec->status != 0 || rec->pad2 != 0) return;  // pad2: unpc_pair_body wrote this packet's PCM
    // element rounds: a packet that ended before this element leaves these channels zero (codec/ALACDecoder.cu:971-998)
    const bool absent = V.elemBit && rec->elementChannels == 0;
    const uint32_t n = absent ? (V.round ? A.numSamplesOut[p] : A.frameSize) : rec->numSamples;
    const uint32_t shb = absent ? 0 : rec->bytesShifted;
    const int32_t mixRes = rec->mixRes, mixBits = rec->mixBits;
    const int32_t *u = V.plane + (uint64_t)p * CH * A.frameSize;
    const int32_t *v = u + A.frameSize;
    constexpr uint32_t BPS = bytes_per_sample(DEPTH);
    const uint32_t och = V.outChannels;
    uint8_t *out = A.pcmOut + ((uint64_t)p * A.frameSize * och + V.outFirst) * BPS;
    if constexpr (DEPTH == 16 && CH == 2) {
        // the common shape (16-bit stereo into a stereo frame): four frames per thread — two 16-byte loads, one 16-byte
        // store instead of 4-byte accesses (the kernel is a 6 GB stream at 125 000 packets)
        if (och == 2 && (A.frameSize & 3) == 0) {
            typedef int32_t I4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
            const uint32_t n4 = n & ~3u;
            for (uint32_t j = (part * blockDim.x + threadIdx.x) * 4; j < n4; j += bx * blockDim.x * 4) {
                const I4 zz = {0, 0, 0, 0};
                const I4 uu = absent ? zz : *(const I4 *)(u + j), vv = absent ? zz : *(const I4 *)(v + j);
                U4 o;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int32_t l, r;
                    if (mixRes != 0) {
                        l = uu[k] + vv[k] - ((mixRes * vv[k]) >> mixBits);
                        r = l - vv[k];
                    } else {
                        l = uu[k];
                        r = vv[k];
                    }
                    o[k] = ((uint32_t)(uint16_t)l) | ((uint32_t)r << 16);
                }
                *(U4 *)(out + (uint64_t)j * 4) = o;
            }
            // the last n mod 4 frames of a short packet
            for (uint32_t j = n4 + part * blockDim.x + threadIdx.x; j < n; j += bx * blockDim.x) {
                const int32_t uu = absent ? 0 : u[j], vv = absent ? 0 : v[j];
                int32_t l, r;
                if (mixRes != 0) {
                    l = uu + vv - ((mixRes * vv) >> mixBits);
                    r = l - vv;
                } else {
                    l = uu;
                    r = vv;
                }
                *(uint32_t *)(out + (uint64_t)j * 4) = ((uint32_t)(uint16_t)l) | ((uint32_t)r << 16);
            }
            return;
        }
    }
    uint32_t done = 0;  // frames [0, done) were written by a vector path
    if constexpr ((DEPTH == 24 || DEPTH == 20) && CH == 2) {
        // 20- / 24-bit stereo into a stereo frame, at most one shifted-off byte per sample (what every encoder emits): four
        // frames per thread — two 16-byte loads, the eight shifted-off bytes as two funnel-shifted words of the staged stream
        // (codec/ALACDecoder.cu:282-338: (x << shift) | shiftUV), 24 bytes out as three 8-byte stores.  Sample by sample
        // (six byte stores and two bit reads per frame) this kernel took 0.41 ms at 10 000 packets, 16-bit: 0.07.
        if (och == 2 && (A.frameSize & 3) == 0 && shb <= 1) {
            typedef int32_t I4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t U2 __attribute__((ext_vector_type(2), aligned(8)));
            const uint32_t n4 = n & ~3u;
            const uint64_t off = A.offsets[p];
            const uint64_t wordBase = off >> 2, bit0 = (off & 3) * 8 + rec->shiftPos;
            for (uint32_t j = (part * blockDim.x + threadIdx.x) * 4; j < n4; j += bx * blockDim.x * 4) {
                const I4 zz = {0, 0, 0, 0};
                const I4 uu = absent ? zz : *(const I4 *)(u + j), vv = absent ? zz : *(const I4 *)(v + j);
                uint32_t x[2] = {0, 0};  // L0 R0 L1 R1 | L2 R2 L3 R3, one byte each, first byte in the top bits
                const bool shifted = DEPTH >= 24 && shb != 0;  // (the reference's 20-bit un-mix takes no shift buffer)
                if (shifted) {
                    const uint64_t b = bit0 + (uint64_t)j * 16;
                    const uint64_t i = min(wordBase + (b >> 5), V.capWords - 3);  // packets of status 0 lie inside the stage
                    const uint32_t sh = (uint32_t)(b & 31);
                    const uint64_t ts = dec_tail_start(V);
                    const uint32_t w0 = dec_word(V, ts, i), w1 = dec_word(V, ts, i + 1), w2 = dec_word(V, ts, i + 2);
                    x[0] = sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
                    x[1] = sh ? (w1 << sh) | (w2 >> (32 - sh)) : w1;
                }
                uint32_t s[8];  // L0 R0 L1 R1 L2 R2 L3 R3 as 24-bit little-endian fields
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int32_t l, r;
                    if (mixRes != 0) {
                        l = uu[k] + vv[k] - ((mixRes * vv[k]) >> mixBits);
                        r = l - vv[k];
                    } else {
                        l = uu[k];
                        r = vv[k];
                    }
                    if (shifted) {
                        const uint32_t xx = x[k >> 1] >> ((k & 1) ? 0 : 16);
                        l = (int32_t)(((uint32_t)l << 8) | ((xx >> 8) & 0xffu));
                        r = (int32_t)(((uint32_t)r << 8) | (xx & 0xffu));
                    }
                    if constexpr (DEPTH == 20) {
                        l = (int32_t)((uint32_t)l << 4);
                        r = (int32_t)((uint32_t)r << 4);
                    }
                    s[2 * k] = (uint32_t)l & 0xffffffu;
                    s[2 * k + 1] = (uint32_t)r & 0xffffffu;
                }
                // four 3-byte fields make three words
                U2 *q = (U2 *)(out + (uint64_t)j * 6);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t a0 = s[4 * h], a1 = s[4 * h + 1], a2 = s[4 * h + 2], a3 = s[4 * h + 3];
                    const uint32_t d0 = a0 | (a1 << 24), d1 = (a1 >> 8) | (a2 << 16), d2 = (a2 >> 16) | (a3 << 8);
                    if (h == 0) {
                        const U2 t0 = {d0, d1};
                        q[0] = t0;
                        s[0] = d2;  // first half of the middle store
                    } else {
                        const U2 t1 = {s[0], d0}, t2 = {d1, d2};
                        q[1] = t1;
                        q[2] = t2;
                    }
                }
            }
            done = n4;
        }
    }
    for (uint32_t j = done + part * blockDim.x + threadIdx.x; j < n; j += bx * blockDim.x) {
        int32_t l, r = 0;
        if constexpr (CH == 2) {
            const int32_t uu = absent ? 0 : u[j], vv = absent ? 0 : v[j];
            if (mixRes != 0) {
                l = uu + vv - ((mixRes * vv) >> mixBits);
                r = l - vv;
            } else {
                l = uu;
                r = vv;
            }
        } else {
            l = absent ? 0 : u[j];
        }
        if (shb != 0 && DEPTH >= 24) {
            const uint8_t *base = A.stream + A.offsets[p];
            const uint64_t nbytes = A.offsets[p + 1] - A.offsets[p];
            uint64_t sp = rec->shiftPos + (uint64_t)j * CH * shb * 8;
            l = (int32_t)(((uint32_t)l << (shb * 8)) | read_bits(base, nbytes, sp, shb * 8));
            if constexpr (CH == 2) r = (int32_t)(((uint32_t)r << (shb * 8)) | read_bits(base, nbytes, sp, shb * 8));
        }
        uint8_t *q = out + (uint64_t)j * och * BPS;
        if (DEPTH == 16 && CH == 2 && och == 2) {
            *(uint32_t *)q = ((uint32_t)(uint16_t)l) | ((uint32_t)r << 16);
        } else {
            put_sample<DEPTH>(q, l);
            if constexpr (CH == 2) put_sample<DEPTH>(q + BPS, r);
        }
    }
}

// Fused launch: workgroup = (packet, part of the frame).  Separate launches: the workgroups walk k_dec_header's list of the
// packets nobody else writes (with pairs and direct uncompressed elements that is none of the benchmark's packets; launching
// a workgroup per packet just to read a record and leave cost 0.5 ms at 125 000 packets).
template <int DEPTH, int CH>
__global__ __launch_bounds__(256) void k_dec_unmix(DecV1Args V)
{
    const uint32_t bx = blocks_per_packet(V.d.frameSize);
    if (!V.lists) {
        unmix_part<DEPTH, CH>(V, blockIdx.x / bx, blockIdx.x % bx, bx);
        return;
    }
    const DecLists L = dec_lists(V);
    const uint64_t work = (uint64_t)L.cnt[5] * bx;
    for (uint64_t i = blockIdx.x; i < work; i += gridDim.x) unmix_part<DEPTH, CH>(V, L.rest[i / bx], (uint32_t)(i % bx), bx);
}

template <int DEPTH>
static void launch_unmix_v1(const DecV1Args &V, hipStream_t st)
{
    const uint64_t all = (uint64_t)blocks_per_packet(V.d.frameSize) * V.d.numPackets;
    dim3 grid((uint32_t)(V.lists && all > 8192 ? 8192 : all));
    if (V.d.numChannels == 2)
        hipLaunchKernelGGL((k_dec_unmix<DEPTH, 2>), grid, dim3(256), 0, st, V);
    else
        hipLaunchKernelGGL((k_dec_unmix<DEPTH, 1>), grid, dim3(256), 0, st, V);
}

// everything after the staging of the stream: one pass of the pipeline over the elements V describes
// side: optional second stream + fork / join events — the clears of the residual plane (328 MB at 10 000 stereo packets: 50 us)
// and of the progress words run there beside k_dec_stage / k_dec_header and are joined in front of the entropy launch
struct DecSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
static hipError_t decode_v1_pass(const DecV1Args &V0, hipStream_t st, const DecSide *side = nullptr, bool stageFirst = false)
{
    DecV1Args V = V0;
    const DecodeArgs &da = V.d;
    const uint64_t planeBytes = (uint64_t)da.numPackets * da.numChannels * da.frameSize * 4;
    const int forced0 = V.d.optFused;
    const bool fused0 = forced0 >= 0 ? forced0 != 0 : dec_fused_auto(da.numPackets, da.numChannels);
    const bool useSide = side && side->stream && fused0;
    hipStream_t sc = useSide ? side->stream : st;  // the clears
    if (useSide) {
        (void)hipEventRecord(side->fork, st);
        (void)hipStreamWaitEvent(sc, side->fork, 0);
    }
    // zero runs only move the index (k_dec_entropy, fused launch) — except in k_dec_entropy_wide, which writes the zeros of its
    // runs itself (in the fused launch the same code made the entropy wave, the launch's serial chain, slower than the 50 us
    // fill it saves: 10 000 packets 1.95 -> 2.14 ms):
    // every sample a later kernel reads is then written by somebody (coded rows: residuals + run zeros; uncompressed rows:
    // k_dec_raw; absent elements of a > 2-channel round: k_dec_unmix reads nothing), and the fill is left out
    const bool zerosWritten = !fused0;
    // (the runtime's fill: a kernel of this library in its place — 44 us for the 328 MB of 10 000 stereo packets, HBM write speed —
    // measured 1.745 against 1.733 ms per pass: no bubble in front of a fill that IS the pass's first real work, unlike the 4-byte
    // ones DecZero replaced)
    if (!zerosWritten) (void)hipMemsetAsync(V.plane, 0, planeBytes, sc);
    if (useSide) (void)hipMemsetAsync(V.prog, 0, (size_t)da.numPackets * 8, sc);
    V.lists = fused0 ? 0u : 1u;
    V.pairs = (!fused0 && V.d.optPair != 0) ? 1u : 0u;
    // the pass's counters: cleared by its first kernel (k_dec_tail / k_dec_stage, see DecZero) where there is one
    DecZero zero;
    zero.a = V.lists ? dec_lists(V).cnt : nullptr;
    zero.nA = kDecCounters;
    zero.b = V.mismatch;
    if (!stageFirst) {
        if (zero.b) (void)hipMemsetAsync(zero.b, 0, 4, st);
        if (zero.a) (void)hipMemsetAsync(zero.a, 0, kDecCounters * 4, st);
    }
    // DIRECT: separate launches of a mono / stereo stream whose buffer is dword aligned (the fused launch keeps the staged copy:
    // there the entropy wave is the launch's serial chain and the 38 us copy is cheaper than a swap per word on that chain)
    // Measured at 125 000 packets (round 4, same box, A/B in one process): 16-bit 5.55 -> 5.39 ms; 24-bit 6.92 -> 7.06 ms — there
    // half of the stream is shifted-off bytes, which the predictor pairs fetch word by word (a swap and a select per word on
    // THEIR chain), so 20- / 24- / 32-bit streams keep the staged copy.
    // ... and only from kDecDirectPackets on (option dec_direct = 1, the default; 2 = whenever legal): with the entropy launch at
    // one wave per SIMD or less every instruction on the lanes' serial chain counts, and the swap per word costs more than the
    // copy it saves — 34 000 / 48 000 / 64 000 / 90 000 packets, direct against staged, one box: 2.99 / 2.85, 3.27 / 3.18,
    // 3.36 / 3.29, 4.25 / 4.30 ms (profiles/r04/decode_direct_sweep.log)
    const bool directWanted = V.d.optDirect == 2 || (V.d.optDirect == 1 && da.numPackets >= kDecDirectPackets);
    const bool direct = stageFirst && !fused0 && V.elemBit == nullptr && ((uintptr_t)da.stream & 3) == 0 && directWanted &&
                        da.bitDepth == 16;
    if (direct) {
        V.raw = (const uint32_t *)da.stream;
        V.tail = V.words;                // the first words of the (otherwise unused) staging area
        V.capWords = 1ull << 40;         // no staging area, no truncation: the word limit is the stream's own end + 64
        hipLaunchKernelGGL(k_dec_tail, dim3(1), dim3(kDecTailWords), 0, st, da.stream, da.offsets, da.numPackets, const_cast<uint32_t *>(V.words),
                           zero);
    } else if (stageFirst)
        hipLaunchKernelGGL(k_dec_stage, dim3(2048), dim3(256), 0, st, da.stream, da.offsets, da.numPackets, const_cast<uint32_t *>(V.words),
                           V.capWords, zero);
    hipLaunchKernelGGL(k_dec_header, dim3((da.numPackets + 64 * kHdrWaves - 1) / (64 * kHdrWaves)), dim3(64 * kHdrWaves), 0, st, V);
    if (useSide) {
        (void)hipEventRecord(side->join, sc);
        (void)hipStreamWaitEvent(st, side->join, 0);
    }
    const uint64_t lanes = (uint64_t)da.numPackets * da.numChannels;
    const uint32_t nEnt = (da.numPackets + 63) / 64;
    // One launch (entropy lanes followed by the predictor waves, producer/consumer through HBM) where the chains are few
    // enough that a stage is as slow as its longest serial chain; separate launches where every kernel fills the machine by
    // itself (no polling, no release fence per publish).  Measured, fused / separate, 16-bit stereo packets (round 3, HEAD;
    // profiles/r03/regime_sweep.log): 10 000 1.72 / 2.9 ms, 22 000 2.19 / 2.78, 26 000 2.58 / 2.86, 30 000 2.78 / 2.85,
    // 34 000 2.84 / 2.88, 125 000 17.6 (round 1) / 5.3.  Option dec_fused (ALAC_HIP_DEC_FUSED) = 0 / 1 forces.
    const int forced = V.d.optFused;
    const bool fused = forced >= 0 ? forced != 0 : dec_fused_auto(da.numPackets, da.numChannels);
    if (fused) {
        const uint32_t nEntWg = (da.numPackets + kFusedPpw - 1) / kFusedPpw;
        hipLaunchKernelGGL(k_dec_fused_wg, dim3(nEntWg + (da.numPackets + 3) / 4), dim3(256), 0, st, V, nEntWg);
    } else {
        hipLaunchKernelGGL(k_dec_raw, dim3(da.numPackets < 4096u ? da.numPackets : 4096u), dim3(256), 0, st, V);
        // deferred residual stores, four 16-byte stores per round of sixteen consecutive residuals (round 2, 4-byte stores:
        // paid only up to two entropy waves per SIMD; with the wide stores, measured whole decode pass at 125 000 / 250 000 /
        // 500 000 packets: 9.31 -> 8.19, 19.4 -> 14.2, 38.1 -> 26.5 ms — the kernel was bound by the number of store
        // instructions whose 64 lanes hit 64 different cache lines, which a CU's address path takes one line at a time)
        if (direct)
            hipLaunchKernelGGL(k_dec_entropy_wide<true>, dim3((nEnt + kEntWavesPerWg - 1) / kEntWavesPerWg), dim3(64 * kEntWavesPerWg), 0, st, V, nEnt);
        else
            hipLaunchKernelGGL(k_dec_entropy_wide<false>, dim3((nEnt + kEntWavesPerWg - 1) / kEntWavesPerWg), dim3(64 * kEntWavesPerWg), 0, st, V, nEnt);
        // chains sorted by tap count, one lane per chain
        // (five lists, each rounded up to whole waves)
        const dim3 ugrid(((uint32_t)((lanes + 63) / 64) + 6 + kEntWavesPerWg - 1) / kEntWavesPerWg), ublock(64 * kEntWavesPerWg);
        if (da.bitDepth == 24) hipLaunchKernelGGL(k_dec_unpc_wide<24>, ugrid, ublock, 0, st, V);
        else if (da.bitDepth == 20) hipLaunchKernelGGL(k_dec_unpc_wide<20>, ugrid, ublock, 0, st, V);
        else hipLaunchKernelGGL(k_dec_unpc_wide<16>, ugrid, ublock, 0, st, V);
    }
    hipLaunchKernelGGL(k_dec_unpc, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, st, V);
    switch (da.bitDepth) {
    case 16: launch_unmix_v1<16>(V, st); break;
    case 20: launch_unmix_v1<20>(V, st); break;
    case 24: launch_unmix_v1<24>(V, st); break;
    case 32: launch_unmix_v1<32>(V, st); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static DecV1Args decode_v1_args(const DecodeArgs &da, uint32_t *words, uint64_t capWords, int32_t *plane, uint32_t *prog)
{
    DecV1Args V;
    V.d = da;
    V.words = words;
    V.raw = nullptr;
    V.tail = nullptr;
    V.capWords = capWords;
    V.plane = plane;
    V.prog = prog;
    // publish every (mask + 1) rounds of 16 residuals: workgroup-local followers 8 (a publish only waits for the wave's own
    // stores; measured at 10 000 packets, mask 31 / 15 / 7 / 3 / 1: 1.749 / 1.723 / 1.717 / 1.717 / 1.729 ms), followers anywhere
    // on the chip 32 (every publish is an L2 write-back)
    V.pubMask = 7u;
    V.elemBit = nullptr;
    V.mismatch = nullptr;
    V.lists = V.pairs = 0;  // decode_v1_pass decides
    V.round = 0;
    V.outChannels = da.numChannels;
    V.outFirst = 0;
    return V;
}

hipError_t launch_decode_v1(const DecodeArgs &da, uint32_t *words, uint64_t capWords, int32_t *plane, uint32_t *prog,
                            hipStream_t st, uint32_t *mismatch, hipStream_t sideStream, hipEvent_t fork, hipEvent_t join)
{
    if (da.numPackets == 0) return hipSuccess;
    DecV1Args V = decode_v1_args(da, words, capWords, plane, prog);
    V.mismatch = mismatch;
    DecSide side;
    side.stream = sideStream;
    side.fork = fork;
    side.join = join;
    return decode_v1_pass(V, st, sideStream ? &side : nullptr, true);
}

// counts the packets whose status is `code` (the element-sequence mismatch of launch_decode_v1_elements)
__global__ void k_dec_count_status(const int32_t *status, uint32_t n, int32_t code, uint32_t *count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && status[i] == code) atomicAdd(count, 1u);
}

hipError_t launch_count_status(const int32_t *status, uint32_t n, int32_t code, uint32_t *count, hipStream_t st)
{
    (void)hipMemsetAsync(count, 0, 4, st);
    hipLaunchKernelGGL(k_dec_count_status, dim3((n + 255) / 256), dim3(256), 0, st, status, n, code, count);
    return hipGetLastError();
}

hipError_t launch_decode_v1_elements(const DecodeArgs &da, const McElement *el, uint32_t numElements, uint32_t *words,
                                     uint64_t capWords, int32_t *plane, uint32_t *prog, uint32_t *elemBit,
                                     uint32_t *mismatch, hipStream_t st)
{
    if (da.numPackets == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dec_stage, dim3(2048), dim3(256), 0, st, da.stream, da.offsets, da.numPackets, words, capWords, DecZero{nullptr, nullptr, 0});
    (void)hipMemsetAsync(elemBit, 0, (size_t)da.numPackets * 4, st);
    (void)hipMemsetAsync(mismatch, 0, 4, st);
    for (uint32_t r = 0; r < numElements; r++) {
        DecodeArgs dr = da;
        dr.numChannels = el[r].channels;  // this round's elements as mono / stereo packets
        DecV1Args V = decode_v1_args(dr, words, capWords, plane, prog);
        V.elemBit = elemBit;
        V.round = r;
        V.outChannels = da.numChannels;
        V.outFirst = el[r].first;
        const hipError_t e = decode_v1_pass(V, st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_dec_count_status, dim3((da.numPackets + 255) / 256), dim3(256), 0, st, da.statusOut, da.numPackets,
                       -4, mismatch);
    return hipGetLastError();
}

}  // namespace alacdev
