// alac_encode.hip — batch ALAC encode kernels for gfx950 (wave64).
//
// Pipeline per call (all on one HIP stream, no host round trips):
//   k_encode_stereo / k_encode_mono   search + final predictor + entropy coder, one lane per
//                                     channel chain; writes per-packet records, per-channel
//                                     bit strings (scratch) and packet sizes
//   k_scan_sizes                      device-wide exclusive scan of the packet sizes
//   k_pack                            one workgroup per packet: header + shift-off bytes + U bits
//                                     + V bits (+ escape payload) funnel-shifted into the packed
//                                     byte-aligned stream at the scanned offset
//
// Reference control flow restated: ALACEncoder::EncodeStereo codec/ALACEncoder.cu:290-558,
// EncodeStereoEscape :749-806, EncodeMono :812-963, Encode :973-1057.
#include <cstdlib>
#include "alac_dev.hpp"
#include "alac_kernels.hpp"

namespace alacdev {

// ------------------------------------------------------------------------------------------------
// predictor passes (lane-serial)
// ------------------------------------------------------------------------------------------------

// One pc_block call (codec/dp_enc.c:77) over a chain, fused with the mix that feeds it.
// `num` is pc_block's num (coefficients adapt for j in (NA, num)); residuals of positions
// j < P are handed to sink(j, del).  P <= max(num, NA+1) by construction of the callers.
template <int DEPTH, int CH, int NA, class Sink>
__device__ __forceinline__ void lms_pass(const uint8_t *pk, int mixres, int ch, uint32_t P,
                                         uint32_t num, int32_t *coefs, uint32_t chanshift,
                                         Sink &&sink)
{
    Lms<NA> s;
#pragma unroll
    for (int k = 0; k < NA; k++) s.a[k] = coefs[k];
#pragma unroll
    for (int k = 0; k <= NA; k++) s.h[k] = 0;

    auto fetch = [&](uint32_t j) -> int32_t {
        if constexpr (CH == 2) {
            int32_t l, r;
            load_lr<DEPTH>(pk, j, l, r);
            return mix_sample(mixres, ch, l, r);
        } else {
            return load_sample<DEPTH>(pk, j) >> (8 * (int)bytes_shifted(DEPTH));
        }
    };

    int32_t xn = P ? fetch(0) : 0;
    for (uint32_t j = 0; j < P; j++) {
        const int32_t x = xn;
        if (j + 1 < P) xn = fetch(j + 1);  // next sample in flight while this step computes
        int32_t del;
        if (j == 0) {
            del = x;
            lms_push<NA>(s, x);
        } else if (j <= (uint32_t)NA) {
            del = sext(x - s.h[0], chanshift);
            lms_push<NA>(s, x);
        } else {
            del = lms_step_enc<NA>(s, x, chanshift);
        }
        sink(j, del);
    }
#pragma unroll
    for (int k = 0; k < NA; k++) coefs[k] = s.a[k];
    (void)num;
}

// ------------------------------------------------------------------------------------------------
// stereo element: one lane per channel (even lane = U, odd lane = V of the same segment)
// ------------------------------------------------------------------------------------------------

template <int DEPTH, int NA>
__device__ __forceinline__ uint32_t numuv_cost(const uint8_t *pk, int best, int ch, uint32_t N,
                                               int32_t *row, uint32_t chanBits,
                                               const int32_t *pred, uint64_t predStride)
{
    // codec/ALACEncoder.cu:420-452: 8 converge passes over N/32, then dyn_comp over N/8 whose
    // tail [max(N/32, NA+1), N/8) still holds the mixRes = 4 search pass
    const uint32_t chanshift = 32 - chanBits;
    const uint32_t n8 = N / 8, n32 = N / 32;
    uint32_t P2 = n32 > (uint32_t)(NA + 1) ? n32 : (uint32_t)(NA + 1);
    P2 = P2 < n8 ? P2 : n8;
    Golomb g;
    gol_reset(g, kMB0, kPB0, kKB0);
    for (int conv = 0; conv < 8; conv++) {
        if (conv < 7) {
            if (n32 > (uint32_t)(NA + 1))
                lms_pass<DEPTH, 2, NA>(pk, best, ch, n32, n32, row, chanshift, [](uint32_t, int32_t) {});
        } else {
            lms_pass<DEPTH, 2, NA>(pk, best, ch, P2, n32, row, chanshift, [&](uint32_t j, int32_t del) {
                gol_sym<false>(g, del, j + 1 == n8, chanBits);
            });
        }
    }
    for (uint32_t j = P2; j < n8; j++) gol_sym<false>(g, pred[(uint64_t)j * predStride], j + 1 == n8, chanBits);
    return g.bits * 8 + 16 * NA;
}

template <int DEPTH, int NA>
__device__ __forceinline__ uint32_t final_pass(const uint8_t *pk, int best, int ch, uint32_t N,
                                               int32_t *row, uint32_t chanBits, uint32_t pb,
                                               uint32_t *words, uint32_t wcap)
{
    // codec/ALACEncoder.cu:505-532 (stereo) / :941-945 (mono)
    Golomb g;
    gol_reset(g, kMB0, pb, kKB0);
    g.wp = words;
    g.wcap = wcap;
    lms_pass<DEPTH, 2, NA>(pk, best, ch, N, N, row, 32 - chanBits, [&](uint32_t j, int32_t del) {
        gol_sym<true>(g, del, j + 1 == N, chanBits);
    });
    gol_flush<true>(g);
    return g.bits;
}

template <int DEPTH>
__global__ __launch_bounds__(64) void k_encode_stereo(EncodeArgs A)
{
    const uint32_t gl = blockIdx.x * 64u + threadIdx.x;
    const uint32_t seg = gl >> 1;
    const int ch = (int)(gl & 1);
    if (seg >= A.numSegments) return;

    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    constexpr uint32_t chanBits = DEPTH - 8 * SHB + 1;  // :334
    constexpr uint32_t chanshift = 32 - chanBits;
    const uint32_t frameBytes = A.frameSize * 2u * bytes_per_sample(DEPTH);

    // persistent rows of this channel: row 3 (4 taps) and row 7 (8 taps)
    int32_t a3[4], a7[8];
    if (A.state && A.stateIn) {
        const int16_t *s = A.state + (uint64_t)seg * 64 + ch * 32;
#pragma unroll
        for (int k = 0; k < 4; k++) a3[k] = s[k];
#pragma unroll
        for (int k = 0; k < 8; k++) a7[k] = s[16 + k];
    } else {  // init_coefs, codec/dp_enc.c:49-60 with DENSHIFT_DEFAULT 9
        a3[0] = a7[0] = (38 * 512) >> 4;
        a3[1] = a7[1] = (-29 * 512) >> 4;
        a3[2] = a7[2] = (-2 * 512) >> 4;
        a3[3] = a7[3] = 0;
#pragma unroll
        for (int k = 4; k < 8; k++) a7[k] = 0;
    }

    const uint32_t p0 = A.segFirst ? A.segFirst[seg] : seg;
    const uint32_t p1 = A.segFirst ? A.segFirst[seg + 1] : seg + 1;
    int32_t *pred = A.pred + gl;  // [j][lane] so a wave's stores coalesce
    const uint64_t predStride = A.predStride;

    for (uint32_t p = p0; p < p1; p++) {
        uint32_t N = A.numSamples ? A.numSamples[p] : A.frameSize;
        N = N < A.frameSize ? N : A.frameSize;
        const uint8_t *pk = A.pcm + (uint64_t)p * frameBytes;
        const uint32_t partial = (N != A.frameSize);
        const uint32_t n8 = N / 8;

        // ---- mixRes search, :353-379: five passes walking row 7 one after the other ----
        int best = 0;
        uint32_t minb = 1u << 31;
        for (int r = 0; r <= kMaxRes; r++) {
            Golomb g;
            gol_reset(g, kMB0, kPB0, kKB0);
            const bool keep = (r == kMaxRes);
            lms_pass<DEPTH, 2, 8>(pk, r, ch, n8, n8, a7, chanshift, [&](uint32_t j, int32_t del) {
                gol_sym<false>(g, del, j + 1 == n8, chanBits);
                if (keep) pred[(uint64_t)j * predStride] = del;
            });
            const uint32_t tot = g.bits + __shfl_xor(g.bits, 1);
            if (tot < minb) {
                minb = tot;
                best = r;
            }
        }

        // ---- numUV search, :418-452 ----
        uint32_t numMine = 4;
        uint32_t minMine = numuv_cost<DEPTH, 4>(pk, best, ch, N, a3, chanBits, pred, predStride);
        {
            const uint32_t c8 = numuv_cost<DEPTH, 8>(pk, best, ch, N, a7, chanBits, pred, predStride);
            if (c8 < minMine) {
                minMine = c8;
                numMine = 8;
            }
        }

        // ---- escape estimate, :455-461 ----
        uint32_t minBits = minMine + __shfl_xor(minMine, 1) + 64 + (partial ? 32 : 0);
        minBits += N * (SHB * 8) * 2;
        const uint32_t escapeBits = N * DEPTH * 2 + (partial ? 32 : 0) + 16;
        bool doEscape = (minBits >= escapeBits);

        // header coefficients are the row as it stands before the final pass, :477-485
        PacketRec *rec = A.recs + p;
        rec->c[ch].num = (uint16_t)numMine;
#pragma unroll
        for (int k = 0; k < 8; k++) rec->c[ch].coefs[k] = (int16_t)(numMine == 4 ? (k < 4 ? a3[k] : 0) : a7[k]);

        uint32_t bits = 0;
        if (!doEscape) {
            uint32_t *words = A.bitWords + ((uint64_t)p * 2 + ch) * A.wcap;
            bits = numMine == 4
                       ? final_pass<DEPTH, 4>(pk, best, ch, N, a3, chanBits, kPB0, words, A.wcap)
                       : final_pass<DEPTH, 8>(pk, best, ch, N, a7, chanBits, kPB0, words, A.wcap);
        }
        rec->c[ch].bits = bits;
        const uint32_t obits = __shfl_xor(bits, 1);
        const uint32_t onum = __shfl_xor(numMine, 1);
        uint32_t body = 0;
        if (!doEscape) {
            // :537-543 compressed element not smaller than the escape element
            body = 12 + 4 + (partial ? 32 : 0) + 16 + (16 + 16 * numMine) + (16 + 16 * onum) +
                   N * (SHB * 8) * 2 + bits + obits;
            if (body >= escapeBits) doEscape = true;
        }
        if (doEscape) body = 12 + 4 + (partial ? 32 : 0) + N * DEPTH * 2;
        if (ch == 0) {
            rec->numSamples = N;
            rec->escape = doEscape ? 1u : 0u;
            rec->mixRes = (uint32_t)best;
            rec->totalBits = 7 + body + 3;
            A.packetBytes[p] = (7 + body + 3 + 7) / 8;
        }
    }

    if (A.state) {
        int16_t *s = A.state + (uint64_t)seg * 64 + ch * 32;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            s[k] = (int16_t)(k < 4 ? a3[k] : 0);
            s[16 + k] = (int16_t)(k < 8 ? a7[k] : 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mono element: one lane per segment (codec/ALACEncoder.cu:812-963)
// ------------------------------------------------------------------------------------------------

template <int DEPTH, int NA>
__device__ __forceinline__ uint32_t mono_cost(const uint8_t *pk, uint32_t N, int32_t *row,
                                              uint32_t chanBits)
{
    // :881-905: 7 passes over N/32, one over N/8, then dyn_comp over N/8
    const uint32_t chanshift = 32 - chanBits;
    const uint32_t n8 = N / 8, n32 = N / 32;
    for (int conv = 0; conv < 7; conv++)
        if (n32 > (uint32_t)(NA + 1))
            lms_pass<DEPTH, 1, NA>(pk, 0, 0, n32, n32, row, chanshift, [](uint32_t, int32_t) {});
    Golomb g;
    gol_reset(g, kMB0, kPB0, kKB0);
    lms_pass<DEPTH, 1, NA>(pk, 0, 0, n8, n8, row, chanshift, [&](uint32_t j, int32_t del) {
        gol_sym<false>(g, del, j + 1 == n8, chanBits);
    });
    return g.bits * 8 + 16 * NA;
}

template <int DEPTH, int NA>
__device__ __forceinline__ uint32_t mono_final(const uint8_t *pk, uint32_t N, int32_t *row,
                                               uint32_t chanBits, uint32_t *words, uint32_t wcap)
{
    Golomb g;
    gol_reset(g, kMB0, kPB0, kKB0);  // set_standard_ag_params, :944
    g.wp = words;
    g.wcap = wcap;
    lms_pass<DEPTH, 1, NA>(pk, 0, 0, N, N, row, 32 - chanBits, [&](uint32_t j, int32_t del) {
        gol_sym<true>(g, del, j + 1 == N, chanBits);
    });
    gol_flush<true>(g);
    return g.bits;
}

template <int DEPTH>
__global__ __launch_bounds__(64) void k_encode_mono(EncodeArgs A)
{
    const uint32_t seg = blockIdx.x * 64u + threadIdx.x;
    if (seg >= A.numSegments) return;

    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    constexpr uint32_t chanBits = DEPTH - 8 * SHB;  // :856
    const uint32_t frameBytes = A.frameSize * bytes_per_sample(DEPTH);

    int32_t a3[4], a7[8];
    if (A.state && A.stateIn) {
        const int16_t *s = A.state + (uint64_t)seg * 64;
#pragma unroll
        for (int k = 0; k < 4; k++) a3[k] = s[k];
#pragma unroll
        for (int k = 0; k < 8; k++) a7[k] = s[16 + k];
    } else {
        a3[0] = a7[0] = (38 * 512) >> 4;
        a3[1] = a7[1] = (-29 * 512) >> 4;
        a3[2] = a7[2] = (-2 * 512) >> 4;
        a3[3] = a7[3] = 0;
#pragma unroll
        for (int k = 4; k < 8; k++) a7[k] = 0;
    }

    const uint32_t p0 = A.segFirst ? A.segFirst[seg] : seg;
    const uint32_t p1 = A.segFirst ? A.segFirst[seg + 1] : seg + 1;

    for (uint32_t p = p0; p < p1; p++) {
        uint32_t N = A.numSamples ? A.numSamples[p] : A.frameSize;
        N = N < A.frameSize ? N : A.frameSize;
        const uint8_t *pk = A.pcm + (uint64_t)p * frameBytes;
        const uint32_t partial = (N != A.frameSize);

        uint32_t bestU = 4;
        uint32_t minBits = mono_cost<DEPTH, 4>(pk, N, a3, chanBits);
        {
            const uint32_t c8 = mono_cost<DEPTH, 8>(pk, N, a7, chanBits);
            if (c8 < minBits) {
                minBits = c8;
                bestU = 8;
            }
        }
        // :907-915
        minBits += 32 + (partial ? 32 : 0) + N * (SHB * 8);
        const uint32_t escapeBits = N * DEPTH + (partial ? 32 : 0) + 16;
        bool doEscape = (minBits >= escapeBits);

        PacketRec *rec = A.recs + p;
        rec->c[0].num = (uint16_t)bestU;
#pragma unroll
        for (int k = 0; k < 8; k++) rec->c[0].coefs[k] = (int16_t)(bestU == 4 ? (k < 4 ? a3[k] : 0) : a7[k]);
        rec->c[1].num = 0;
        rec->c[1].bits = 0;

        uint32_t bits = 0, body;
        if (!doEscape) {
            uint32_t *words = A.bitWords + (uint64_t)p * 2 * A.wcap;
            bits = bestU == 4 ? mono_final<DEPTH, 4>(pk, N, a3, chanBits, words, A.wcap)
                              : mono_final<DEPTH, 8>(pk, N, a7, chanBits, words, A.wcap);
            body = 12 + 4 + (partial ? 32 : 0) + 16 + (16 + 16 * bestU) + N * (SHB * 8) + bits;
            if (body >= escapeBits) doEscape = true;  // :952-958
        }
        if (doEscape) body = 12 + 4 + (partial ? 32 : 0) + N * DEPTH;
        rec->c[0].bits = bits;
        rec->numSamples = N;
        rec->escape = doEscape ? 1u : 0u;
        rec->mixRes = 0;
        rec->totalBits = 7 + body + 3;
        A.packetBytes[p] = (7 + body + 3 + 7) / 8;
    }

    if (A.state) {
        int16_t *s = A.state + (uint64_t)seg * 64;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            s[k] = (int16_t)(k < 4 ? a3[k] : 0);
            s[16 + k] = (int16_t)(k < 8 ? a7[k] : 0);
            if (!A.stateIn) {  // V rows are never touched by a mono element: leave them as init_coefs
                const int16_t init = (int16_t)(k == 0 ? 1216 : k == 1 ? -928 : k == 2 ? -64 : 0);
                s[32 + k] = init;
                s[48 + k] = init;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of packet sizes (uint32 -> uint64 offsets, offsets[n] = total)
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void k_scan_sizes(const uint32_t *sizes, uint64_t *offsets, uint32_t n)
{
    __shared__ uint64_t waveSum[16];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint64_t v = i < n ? sizes[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(incl, d);
            if ((int)lane >= d) incl += t;
        }
        if (lane == 63) waveSum[wave] = incl;
        __syncthreads();
        uint64_t wpre = 0;
        for (uint32_t w = 0; w < wave; w++) wpre += waveSum[w];
        const uint64_t c = carry;
        if (i < n) offsets[i] = c + wpre + incl - v;
        __syncthreads();
        if (tid == 1023) carry = c + wpre + incl;
        __syncthreads();
    }
    if (tid == 0) offsets[n] = carry;
}

// ------------------------------------------------------------------------------------------------
// pack: one workgroup per packet
// ------------------------------------------------------------------------------------------------

// 32 bits of a bit string stored as MSB-first 32-bit words, starting at bit s0 (0 <= s0 < len),
// zero beyond len
__device__ __forceinline__ uint32_t words_fetch32(const uint32_t *w, uint32_t len, uint32_t s0)
{
    const uint32_t i = s0 >> 5, sh = s0 & 31;
    const uint32_t nw = (len + 31) >> 5;
    const uint32_t a = w[i];
    const uint32_t b = (i + 1 < nw) ? w[i + 1] : 0u;
    uint32_t v = sh ? ((a << sh) | (b >> (32 - sh))) : a;
    const uint32_t rem = len - s0;
    if (rem < 32) v &= ~0u << (32 - rem);
    return v;
}

// 32 bits of the string "field(t) for t = 0..count-1", each W bits MSB-first, starting at bit s0
template <int W, class Field>
__device__ __forceinline__ uint32_t fields_fetch32(Field &&field, uint32_t count, uint64_t s0)
{
    uint32_t out = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const uint64_t s = s0 + 16u * half;
        uint32_t t = (uint32_t)(s / W);
        const uint32_t o = (uint32_t)(s % W);
        uint64_t acc = 0;
        uint32_t have = 0;
        while (have < o + 16) {
            const uint64_t f = t < count ? (uint64_t)field(t) : 0ull;
            acc = (W == 32) ? ((acc << 32) | f) : ((acc << W) | (f & ((1ull << W) - 1)));
            have += W;
            t++;
        }
        out = (out << 16) | (uint32_t)((acc >> (have - o - 16)) & 0xffffu);
    }
    return out;
}

template <class Fetch>
__device__ __forceinline__ uint32_t take(Fetch &&fetch32, uint64_t len, int64_t s)
{
    if (s >= (int64_t)len || s <= -32) return 0;
    if (s >= 0) return fetch32((uint64_t)s);
    return fetch32(0) >> (uint32_t)(-s);
}

// MSB-first bit writer over zero-initialised 32-bit words, one word-sized step per field (the header is built by
// one lane while the rest of the workgroup waits: keep it short)
struct HdrWriter {
    uint32_t *w;
    uint32_t pos;
    __device__ __forceinline__ void put(uint32_t v, uint32_t n)
    {
        // n <= 32; v confined to n bits by the callers
        const uint32_t i = pos >> 5, o = pos & 31;
        const uint64_t x = ((uint64_t)v << (64 - n)) >> o;  // field left-aligned at bit o of a 64-bit window
        w[i] |= (uint32_t)(x >> 32);
        if (o + n > 32) w[i + 1] |= (uint32_t)x;
        pos += n;
    }
};

template <int DEPTH, int CH>
__global__ __launch_bounds__(256) void k_pack(PackArgs A)
{
    __shared__ uint32_t hdr[16];
    const uint32_t p = blockIdx.x;
    const PacketRec rec = A.recs[p];
    const uint32_t N = rec.numSamples;
    const uint32_t partial = (N != A.frameSize);
    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    const uint8_t *pk = A.pcm + (uint64_t)p * A.frameSize * CH * bytes_per_sample(DEPTH);

    if (threadIdx.x < 16) hdr[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        // Encode() :989-991 / :1011-1012 element tag, then the element header
        HdrWriter h{hdr, 0};
        h.put(CH == 2 ? 1u : 0u, 3);
        h.put(0, 4);
        h.put(0, 12);
        if (rec.escape) {
            h.put((partial << 3) | 1u, 4);  // :762
            if (partial) h.put(N, 32);
        } else {
            h.put((partial << 3) | (SHB << 1), 4);  // :467 / :921
            if (partial) h.put(N, 32);
            if (CH == 2) {
                h.put((uint32_t)kMixBits, 8);
                h.put(rec.mixRes, 8);
            } else {
                h.put(0, 16);
            }
            for (int c = 0; c < CH; c++) {
                h.put((0u << 4) | kDenShift, 8);
                h.put((4u << 5) | rec.c[c].num, 8);
                for (uint32_t k = 0; k < rec.c[c].num; k++) h.put((uint16_t)rec.c[c].coefs[k], 16);
            }
        }
    }
    // the header LENGTH follows from the record alone, so the bulk copies below need not wait for its bits
    uint32_t hb = 3 + 4 + 12 + 4 + (partial ? 32u : 0u);
    if (!rec.escape) {
        hb += 16;
        for (int c = 0; c < CH; c++) hb += 16 + 16 * rec.c[c].num;
    }
    const uint64_t lenHdr = hb;
    const uint64_t lenShift = rec.escape ? 0 : (uint64_t)N * CH * SHB * 8;
    const uint64_t lenU = rec.escape ? 0 : rec.c[0].bits;
    const uint64_t lenV = (rec.escape || CH == 1) ? 0 : rec.c[1].bits;
    const uint64_t lenRaw = rec.escape ? (uint64_t)N * CH * DEPTH : 0;
    const int64_t offShift = (int64_t)lenHdr;
    const int64_t offU = offShift + (int64_t)lenShift;
    const int64_t offV = offU + (int64_t)lenU;
    const int64_t offRaw = offV + (int64_t)lenV;
    const int64_t offEnd = offRaw + (int64_t)lenRaw;  // ID_END '111', :1036
    const uint32_t nb = (uint32_t)((offEnd + 3 + 7) / 8);

    const uint32_t *wU = A.bitWords + (uint64_t)p * 2 * A.wcap;
    const uint32_t *wV = wU + A.wcap;
    const uint64_t outOff = A.offsets[p];
    const uint32_t mis = (uint32_t)(outOff & 15);  // work in words of a 16-byte aligned frame: 16-byte stores
    uint8_t *outAligned = A.out + (outOff - mis);
    const uint32_t nwords = (mis + nb + 3) / 4;

    auto sampleField = [&](uint32_t t) -> uint32_t { return (uint32_t)load_sample<DEPTH>(pk, t); };

    // Bulk of a packet = the two channel bit strings (or the raw samples of an escape packet): output words that
    // lie completely inside one of them are a plain funnel-shifted copy with a constant shift — no per-word
    // segment search, independent loads.  Words touching a segment boundary (header, U|V seam, tail; a dozen per
    // packet) and the 24-bit shift bytes take the general path below.
    struct Span {
        uint32_t w0, w1;  // aligned output words [w0, w1) fully inside the segment
    };
    auto span_of = [&](int64_t off, uint64_t len) {
        const int64_t base = off + (int64_t)mis * 8;  // bit position from the aligned output start
        Span sp;
        sp.w0 = (uint32_t)((base + 31) >> 5);
        sp.w1 = (uint32_t)((base + (int64_t)len) >> 5);
        if (sp.w1 < sp.w0) sp.w1 = sp.w0;
        return sp;
    };
    auto copy_span = [&](const Span &sp, int64_t off, auto &&srcWord) {
        const uint32_t sh = (uint32_t)((int64_t)sp.w0 * 32 - (off + (int64_t)mis * 8));  // 0..31: source bit of word w0
        auto one = [&](uint32_t w) {
            const uint32_t i = w - sp.w0;
            const uint32_t a = srcWord(i);
            const uint32_t b = sh ? srcWord(i + 1) : 0u;  // exists: the output word ends inside the segment
            const uint32_t v = sh ? __builtin_amdgcn_alignbit(a, b, 32 - sh) : a;
            *(uint32_t *)(outAligned + (uint64_t)w * 4) = __builtin_bswap32(v);
        };
        // groups of four words = one 16-byte store, five independent loads in flight per lane (the copy is
        // latency bound: bytes in flight per lane are what sets its rate)
        const uint32_t g0 = (sp.w0 + 3) & ~3u, g1 = sp.w1 & ~3u;
        if (g0 < g1) {
            for (uint32_t w = sp.w0 + threadIdx.x; w < g0; w += blockDim.x) one(w);
            for (uint32_t w = g0 + 4 * threadIdx.x; w < g1; w += 4 * blockDim.x) {
                const uint32_t i = w - sp.w0;
                uint32_t a[5];
#pragma unroll
                for (int q = 0; q < 5; q++) a[q] = (q < 4 || sh) ? srcWord(i + q) : 0u;
                uint4 v;
                v.x = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[0], a[1], 32 - sh) : a[0]);
                v.y = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[1], a[2], 32 - sh) : a[1]);
                v.z = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[2], a[3], 32 - sh) : a[2]);
                v.w = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[3], a[4], 32 - sh) : a[3]);
                *(uint4 *)(outAligned + (uint64_t)w * 4) = v;
            }
            for (uint32_t w = g1 + threadIdx.x; w < sp.w1; w += blockDim.x) one(w);
        } else {
            for (uint32_t w = sp.w0 + threadIdx.x; w < sp.w1; w += blockDim.x) one(w);
        }
    };
    // Two spans at once (the U and the V bit string of a stereo packet): the loads of BOTH are issued before either is
    // stored, so the packet pays one memory round trip for its bulk instead of two — the packer is bound by the chain of
    // dependent round trips per workgroup (record -> spans -> seams), not by bandwidth.
    auto copy_two = [&](const Span &sa, int64_t offA, auto &&srcA, const Span &sb, int64_t offB, auto &&srcB) {
        const uint32_t shA = (uint32_t)((int64_t)sa.w0 * 32 - (offA + (int64_t)mis * 8));
        const uint32_t shB = (uint32_t)((int64_t)sb.w0 * 32 - (offB + (int64_t)mis * 8));
        auto oneOf = [&](const Span &sp, uint32_t sh, auto &&src, uint32_t w) {
            const uint32_t i = w - sp.w0;
            const uint32_t a = src(i);
            const uint32_t b = sh ? src(i + 1) : 0u;
            *(uint32_t *)(outAligned + (uint64_t)w * 4) = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a, b, 32 - sh) : a);
        };
        const uint32_t ga0 = min((sa.w0 + 3) & ~3u, sa.w1), ga1 = max(sa.w1 & ~3u, ga0);
        const uint32_t gb0 = min((sb.w0 + 3) & ~3u, sb.w1), gb1 = max(sb.w1 & ~3u, gb0);
        const uint32_t na = (ga1 - ga0) / 4, nb2 = (gb1 - gb0) / 4;
        for (uint32_t g = threadIdx.x; g < max(na, nb2); g += blockDim.x) {
            const bool ha = g < na, hb = g < nb2;
            const uint32_t wa = ga0 + 4 * g, wb = gb0 + 4 * g;
            const uint32_t ia = wa - sa.w0, ib = wb - sb.w0;
            uint32_t a[5], b[5];
#pragma unroll
            for (int q = 0; q < 5; q++) a[q] = (ha && (q < 4 || shA)) ? srcA(ia + q) : 0u;
#pragma unroll
            for (int q = 0; q < 5; q++) b[q] = (hb && (q < 4 || shB)) ? srcB(ib + q) : 0u;
            if (ha) {
                uint4 v;
                v.x = __builtin_bswap32(shA ? __builtin_amdgcn_alignbit(a[0], a[1], 32 - shA) : a[0]);
                v.y = __builtin_bswap32(shA ? __builtin_amdgcn_alignbit(a[1], a[2], 32 - shA) : a[1]);
                v.z = __builtin_bswap32(shA ? __builtin_amdgcn_alignbit(a[2], a[3], 32 - shA) : a[2]);
                v.w = __builtin_bswap32(shA ? __builtin_amdgcn_alignbit(a[3], a[4], 32 - shA) : a[3]);
                *(uint4 *)(outAligned + (uint64_t)wa * 4) = v;
            }
            if (hb) {
                uint4 v;
                v.x = __builtin_bswap32(shB ? __builtin_amdgcn_alignbit(b[0], b[1], 32 - shB) : b[0]);
                v.y = __builtin_bswap32(shB ? __builtin_amdgcn_alignbit(b[1], b[2], 32 - shB) : b[1]);
                v.z = __builtin_bswap32(shB ? __builtin_amdgcn_alignbit(b[2], b[3], 32 - shB) : b[2]);
                v.w = __builtin_bswap32(shB ? __builtin_amdgcn_alignbit(b[3], b[4], 32 - shB) : b[3]);
                *(uint4 *)(outAligned + (uint64_t)wb * 4) = v;
            }
        }
        // the few words in front of / behind the 16-byte groups
        for (uint32_t w = sa.w0 + threadIdx.x; w < ga0; w += blockDim.x) oneOf(sa, shA, srcA, w);
        for (uint32_t w = ga1 + threadIdx.x; w < sa.w1; w += blockDim.x) oneOf(sa, shA, srcA, w);
        for (uint32_t w = sb.w0 + threadIdx.x; w < gb0; w += blockDim.x) oneOf(sb, shB, srcB, w);
        for (uint32_t w = gb1 + threadIdx.x; w < sb.w1; w += blockDim.x) oneOf(sb, shB, srcB, w);
    };
    Span spU = {0, 0}, spV = {0, 0}, spR = {0, 0}, spS = {0, 0};
    if (!rec.escape) {
        if constexpr (SHB != 0 && (DEPTH == 24 || DEPTH == 32)) {
            // shift-off bytes (codec/ALACEncoder.cu:489-503): the low SHB bytes of every channel sample, MSB first.
            // 24-bit: one byte per 3-byte sample, four samples (three PCM dwords) per output word;
            // 32-bit: two bytes per sample, two samples (two PCM dwords) per output word.
            spS = span_of(offShift, lenShift);
            const uint32_t *pw = (const uint32_t *)pk;  // packets are dword aligned for these depths (frame * 6 or 8 bytes)
            const uint32_t fields = N * CH;             // samples; the last word may hold fewer than a full group
            copy_span(spS, offShift, [&](uint32_t i) -> uint32_t {
                if constexpr (DEPTH == 24) {
                    const uint32_t t = 4 * i;
                    if (t + 4 <= fields) {
                        const uint32_t w0 = pw[3 * i], w1 = pw[3 * i + 1], w2 = pw[3 * i + 2];
                        return (w0 << 24) | ((w0 >> 24) << 16) | (((w1 >> 16) & 0xffu) << 8) | ((w2 >> 8) & 0xffu);
                    }
                    uint32_t v = 0;
                    for (uint32_t q = 0; q < 4; q++)
                        v = (v << 8) | (t + q < fields ? ((uint32_t)load_sample<24>(pk, t + q) & 0xffu) : 0u);
                    return v;
                } else {
                    const uint32_t t = 2 * i;
                    const uint32_t a = t < fields ? pw[t] : 0u, b = t + 1 < fields ? pw[t + 1] : 0u;
                    return (a << 16) | (b & 0xffffu);
                }
            });
        }
        spU = span_of(offU, lenU);
        if constexpr (CH == 2) {
            spV = span_of(offV, lenV);
            copy_two(spU, offU, [&](uint32_t i) { return wU[i]; }, spV, offV, [&](uint32_t i) { return wV[i]; });
        } else {
            copy_span(spU, offU, [&](uint32_t i) { return wU[i]; });
        }
    } else if constexpr (DEPTH == 16) {
        // raw 16-bit samples, MSB first: two little-endian samples per PCM word, swapped into place by a rotate
        spR = span_of(offRaw, lenRaw);
        const uint32_t *pw = (const uint32_t *)pk;  // 16-byte aligned packet (checked by the C entry point)
        copy_span(spR, offRaw, [&](uint32_t i) {
            const uint32_t x = pw[i];
            return (x << 16) | (x >> 16);
        });
    } else if constexpr (DEPTH == 32) {
        // raw 32-bit samples, MSB first: the word IS the little-endian sample
        spR = span_of(offRaw, lenRaw);
        const uint32_t *pw = (const uint32_t *)pk;
        copy_span(spR, offRaw, [&](uint32_t i) { return pw[i]; });
    } else if constexpr (DEPTH == 24) {
        // raw 24-bit samples, MSB first: four samples (PCM bytes B0..B11 in three dwords) make three words
        // B2 B1 B0 B5 | B4 B3 B8 B7 | B6 B11 B10 B9
        spR = span_of(offRaw, lenRaw);
        const uint32_t *pw = (const uint32_t *)pk;
        const uint32_t ndw = (N * CH * 3 + 3) / 4;  // dwords that hold samples (the tail of the last one is padding)
        copy_span(spR, offRaw, [&](uint32_t i) -> uint32_t {
            const uint32_t g = i / 3, r = i - 3 * g;
            const uint32_t d0 = 3 * g < ndw ? pw[3 * g] : 0u, d1 = 3 * g + 1 < ndw ? pw[3 * g + 1] : 0u,
                           d2 = 3 * g + 2 < ndw ? pw[3 * g + 2] : 0u;
            if (r == 0) return ((d0 << 8) & 0xff000000u) | ((d0 << 8) & 0x00ff0000u) | ((d0 << 8) & 0x0000ff00u) | ((d1 >> 8) & 0xffu);
            if (r == 1) return (d1 << 24) | ((d0 >> 24) << 16) | ((d2 & 0xffu) << 8) | (d1 >> 24);
            return (((d1 >> 16) & 0xffu) << 24) | ((d2 >> 24) << 16) | (((d2 >> 16) & 0xffu) << 8) | ((d2 >> 8) & 0xffu);
        });
    }

    __syncthreads();  // header words complete
    for (uint32_t aw = threadIdx.x; aw < nwords; aw += blockDim.x) {
        if ((aw >= spU.w0 && aw < spU.w1) || (aw >= spV.w0 && aw < spV.w1) || (aw >= spR.w0 && aw < spR.w1) ||
            (aw >= spS.w0 && aw < spS.w1))
            continue;
        const int64_t bp = (int64_t)aw * 32 - (int64_t)mis * 8;
        uint32_t v = take([&](uint64_t s0) { return words_fetch32(hdr, (uint32_t)lenHdr, (uint32_t)s0); }, lenHdr, bp);
        if (!rec.escape) {
            if constexpr (SHB != 0)
                v |= take([&](uint64_t s0) { return fields_fetch32<SHB * 8>(sampleField, N * CH, s0); }, lenShift, bp - offShift);
            v |= take([&](uint64_t s0) { return words_fetch32(wU, (uint32_t)lenU, (uint32_t)s0); }, lenU, bp - offU);
            if constexpr (CH == 2)
                v |= take([&](uint64_t s0) { return words_fetch32(wV, (uint32_t)lenV, (uint32_t)s0); }, lenV, bp - offV);
        } else {
            v |= take([&](uint64_t s0) { return fields_fetch32<DEPTH>(sampleField, N * CH, s0); }, lenRaw, bp - offRaw);
        }
        v |= take([](uint64_t s0) { return 0xE0000000u << (uint32_t)s0; }, 3, bp - offEnd);

        const int64_t b0 = (int64_t)aw * 4 - mis;  // packet byte index of this word's first byte
        if (b0 >= 0 && b0 + 4 <= (int64_t)nb) {
            *(uint32_t *)(outAligned + (uint64_t)aw * 4) = __builtin_bswap32(v);
        } else {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int64_t pb = b0 + b;
                if (pb >= 0 && pb < (int64_t)nb) outAligned[(uint64_t)aw * 4 + b] = (uint8_t)(v >> (24 - 8 * b));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

template <int DEPTH>
static void launch_scan_pack_depth(uint32_t channels, uint32_t *packetBytes, const PackArgs &pa, uint32_t numPackets,
                                   hipStream_t st, hipEvent_t *ev, bool recordScan = true)
{
    if (ev && recordScan) (void)hipEventRecord(ev[kStageScan], st);
    hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(1024), 0, st, (const uint32_t *)packetBytes, (uint64_t *)pa.offsets,
                       numPackets);
    if (ev) (void)hipEventRecord(ev[kStagePack], st);
    // 256 threads per packet.  Measured at 10 000 16-bit / 24-bit packets and at 125 000: 64 threads 0.127 / 0.43 / 1.32 ms,
    // 128: 0.102 / 0.28 / 0.99, 256: 0.100 / 0.234 / 1.06; 512 (which needs __launch_bounds__(512) and its tighter register
    // budget) 0.16 / 0.28 / 1.83 — the copy is bound by the latency of a workgroup's dependent loads, not by bandwidth.
    constexpr uint32_t tpb = 256;
    if (channels == 2)
        hipLaunchKernelGGL((k_pack<DEPTH, 2>), dim3(numPackets), dim3(tpb), 0, st, pa);
    else
        hipLaunchKernelGGL((k_pack<DEPTH, 1>), dim3(numPackets), dim3(tpb), 0, st, pa);
    if (ev) (void)hipEventRecord(ev[kNumStages], st);
}

void launch_scan_sizes(const uint32_t *sizes, uint64_t *offsets, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(1024), 0, st, sizes, offsets, n);
}

void launch_scan_pack(uint32_t depth, uint32_t channels, uint32_t *packetBytes, const PackArgs &pa, uint32_t numPackets,
                      hipStream_t st, hipEvent_t *ev, bool recordScan)
{
    switch (depth) {
    case 16: launch_scan_pack_depth<16>(channels, packetBytes, pa, numPackets, st, ev, recordScan); break;
    case 20: launch_scan_pack_depth<20>(channels, packetBytes, pa, numPackets, st, ev, recordScan); break;
    case 24: launch_scan_pack_depth<24>(channels, packetBytes, pa, numPackets, st, ev, recordScan); break;
    default: launch_scan_pack_depth<32>(channels, packetBytes, pa, numPackets, st, ev, recordScan); break;
    }
}

template <int DEPTH>
static void launch_encode_depth(const EncodeArgs &ea, const PackArgs &pa, uint32_t channels,
                                uint32_t numPackets, hipStream_t st, hipEvent_t *ev)
{
    if (ev)
        for (int i = 0; i <= kStageLms3; i++) (void)hipEventRecord(ev[i], st);
    if (channels == 2) {
        const uint32_t lanes = ea.numSegments * 2;
        hipLaunchKernelGGL(k_encode_stereo<DEPTH>, dim3((lanes + 63) / 64), dim3(64), 0, st, ea);
    } else {
        hipLaunchKernelGGL(k_encode_mono<DEPTH>, dim3((ea.numSegments + 63) / 64), dim3(64), 0, st, ea);
    }
    if (ev) (void)hipEventRecord(ev[kStageGol3], st);
    launch_scan_pack_depth<DEPTH>(channels, ea.packetBytes, pa, numPackets, st, ev);
}

hipError_t launch_encode(uint32_t depth, uint32_t channels, const EncodeArgs &ea, const PackArgs &pa,
                         uint32_t numPackets, hipStream_t st, hipEvent_t *ev)
{
    switch (depth) {
    case 16: launch_encode_depth<16>(ea, pa, channels, numPackets, st, ev); break;
    case 20: launch_encode_depth<20>(ea, pa, channels, numPackets, st, ev); break;
    case 24: launch_encode_depth<24>(ea, pa, channels, numPackets, st, ev); break;
    case 32: launch_encode_depth<32>(ea, pa, channels, numPackets, st, ev); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace alacdev
