// alac_encode.hip — batch ALAC encode kernels for gfx950 (wave64).
//
// Pipeline per call (all on one HIP stream, no host round trips):
//   k_encode_stereo / k_encode_mono   search + final predictor + entropy coder, one lane per
//                                     channel chain; writes per-packet records, per-channel
//                                     bit strings (scratch) and packet sizes
//   k_scan_sizes                      device-wide exclusive scan of the packet sizes
//   k_pack                            one workgroup per packet: header + shift-off bytes + U bits
//                                     + V bits (+ escape payload) funnel-shifted into the packed
//                                     byte-aligned stream at the scanned offset; every wave issues all
//                                     its loads before its first store (one memory round trip)
//
// Reference control flow restated: ALACEncoder::EncodeStereo codec/ALACEncoder.cu:290-558,
// EncodeStereoEscape :749-806, EncodeMono :812-963, Encode :973-1057.
#include <cstdlib>
#include <type_traits>
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

    uint32_t p0 = A.segFirst ? A.segFirst[seg] : seg;
    const uint32_t p1 = A.segFirst ? A.segFirst[seg + 1] : seg + 1;
    if (p1 < p0 || p1 > A.numPackets || p1 - p0 > A.segMax) p0 = p1;  // unvalidated table (EncodeArgs::segMax): no packets
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

    uint32_t p0 = A.segFirst ? A.segFirst[seg] : seg;
    const uint32_t p1 = A.segFirst ? A.segFirst[seg + 1] : seg + 1;
    if (p1 < p0 || p1 > A.numPackets || p1 - p0 > A.segMax) p0 = p1;  // unvalidated table (EncodeArgs::segMax): no packets

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

// Workgroup b owns elements [1024 b, 1024 b + 1024).  It first adds up everything in front of its block on its own —
// thread t sums sizes[t], sizes[t + 1024], ... (coalesced, independent loads; the sizes sit in L2) — instead of waiting
// for its predecessors: no scratch memory, no inter-workgroup hand-off, one launch.  The redundant reads are b * 4 KB per
// workgroup (125 000 packets: 30 MB in all); the single workgroup this replaces took 0.17 ms there.
__global__ __launch_bounds__(1024) void k_scan_sizes(const uint32_t *sizes, uint64_t *offsets, uint32_t n, const uint32_t *segBad)
{
    // a segment table refused on the device (EncodeArgs::segBad): the sizes of packets nobody encoded are whatever the
    // caller's buffer held — every offset is 0 then and k_pack writes nothing
    const bool refused = segBad && *segBad != 0;
    __shared__ uint64_t waveSum[16];
    __shared__ uint64_t wavePre[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = blockIdx.x * 1024u;
    uint64_t before = 0;
    {
        // four sizes per load, four loads in flight: the last block of 125 000 packets re-adds 122 K elements, 30 iterations per
        // thread instead of 122 dependent-latency-bound ones (38 -> ~12 us; base is a multiple of 1024)
        typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
        const U4 *s4 = (const U4 *)sizes;
#pragma unroll 4
        for (uint32_t i = tid; i < base / 4; i += 1024) {
            const U4 q = s4[i];
            before += refused ? 0ull : (uint64_t)q.x + q.y + q.z + q.w;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    const uint32_t i = base + tid;
    const uint64_t v = (i < n && !refused) ? sizes[i] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63) waveSum[wave] = incl;
    if (lane == 0) wavePre[wave] = before;
    __syncthreads();
    uint64_t pre = 0;
    for (uint32_t w = 0; w < 16; w++) pre += wavePre[w] + (w < wave ? waveSum[w] : 0);
    if (i < n) offsets[i] = pre + incl - v;
    if (i == n - 1) offsets[n] = pre + incl;
    if (n == 0 && i == 0) offsets[0] = 0;
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

template <int DEPTH, int CH>
__global__ __launch_bounds__(256) void k_pack(PackArgs A, uint32_t numPackets)
{
    // A workgroup walks packets blockIdx.x, blockIdx.x + gridDim.x, ...: with one workgroup per packet the launch was
    // bound by workgroup DISPATCH (125 000 empty 256-thread workgroups alone take 0.6 ms, measured), not by the copy.
    // Nothing in the loop synchronises the workgroup: waves 0..2 only copy, the last wave builds the header (its lanes
    // are the header's fields), writes the few words around the spans, then joins the copy.
    __shared__ uint32_t hdr[16];  // touched by the last wave only: LDS operations of one wave execute in order
    const uint32_t lane = threadIdx.x & 63;
    const bool seamWave = (threadIdx.x >> 6) == (blockDim.x >> 6) - 1;
    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    const uint32_t refused = A.segBad ? *A.segBad : 0u;  // (requested together with the first record: no extra round trip)
    PacketRec recNext = A.recs[blockIdx.x];
    uint64_t offNext = A.offsets[blockIdx.x];
    if (refused) return;  // the segment table was refused on the device: records and sizes are not to be trusted
    for (uint32_t p = blockIdx.x; p < numPackets; p += gridDim.x) {
    // this packet's record and offset were requested one packet ago
    const PacketRec rec = recNext;
    const uint64_t outOff = offNext;
    if (p + gridDim.x < numPackets) {
        recNext = A.recs[p + gridDim.x];
        offNext = A.offsets[p + gridDim.x];
    }
    const uint32_t N = rec.numSamples;
    const uint32_t partial = (N != A.frameSize);
    const uint8_t *pk = A.pcm + (uint64_t)p * A.frameSize * CH * bytes_per_sample(DEPTH);
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
    const uint32_t mis = (uint32_t)(outOff & 15);  // work in words of a 16-byte aligned frame: 16-byte stores
    uint8_t *outAligned = A.out + (outOff - mis);
    const uint32_t nwords = (mis + nb + 3) / 4;

    auto sampleField = [&](uint32_t t) -> uint32_t { return (uint32_t)load_sample<DEPTH>(pk, t); };

    // Bulk of a packet = the two channel bit strings (or the raw samples of an escape packet): the 16-byte groups of
    // output words that lie completely inside one of them are a plain funnel-shifted copy with a constant shift — no
    // per-word segment search, independent loads, ONE memory round trip per wave.  Every other word (header, up to three
    // words either side of a span, the tail) is a seam word of the last wave, below.
    struct Span {
        uint32_t w0, w1;  // aligned output words [w0, w1), both multiples of four, fully inside the segment
    };
    auto span_of = [&](int64_t off, uint64_t len) {
        const int64_t base = off + (int64_t)mis * 8;  // bit position from the aligned output start
        Span sp;
        sp.w0 = ((uint32_t)((base + 31) >> 5) + 3) & ~3u;
        sp.w1 = (uint32_t)((base + (int64_t)len) >> 5) & ~3u;
        if (sp.w1 < sp.w0) sp.w1 = sp.w0;
        return sp;
    };
    // output word w of a span = source bits [32 (w - w0) + first, + 32) of its segment
    auto first_bit = [&](const Span &sp, int64_t off) { return (uint32_t)((int64_t)sp.w0 * 32 - (off + (int64_t)mis * 8)); };
    auto load5 = [&](uint32_t a[5], bool have, uint32_t i, uint32_t sh, auto &&src) {
        if constexpr (std::is_pointer_v<std::decay_t<decltype(src)>>) {
            // a word string in memory: one 16-byte load (the address is only dword aligned, which global loads
            // allow) + one dword instead of five dword loads — a third of the cache accesses
            typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
            const char *at = (const char *)src + i * 4u;  // 32-bit offset from a uniform base
            U4 v = {0, 0, 0, 0};
            uint32_t e = 0;
            if (have) {
                v = *(const U4 *)at;
                if (sh) e = *(const uint32_t *)(at + 16);
            }
            a[0] = v.x, a[1] = v.y, a[2] = v.z, a[3] = v.w, a[4] = e;
        } else {
#pragma unroll
            for (int q = 0; q < 5; q++) a[q] = (have && (q < 4 || sh)) ? src(i + q) : 0u;
        }
    };
    auto store4 = [&](const uint32_t a[5], uint32_t sh, uint32_t w) {
        uint4 v;
        v.x = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[0], a[1], 32 - sh) : a[0]);
        v.y = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[1], a[2], 32 - sh) : a[1]);
        v.z = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[2], a[3], 32 - sh) : a[2]);
        v.w = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(a[3], a[4], 32 - sh) : a[3]);
        *(uint4 *)(outAligned + (uint64_t)w * 4) = v;
    };
    auto copy_span = [&](const Span &sp, int64_t off, auto &&srcWord) {
        const uint32_t fb = first_bit(sp, off), i0 = fb >> 5, sh = fb & 31;
        // five independent loads in flight per lane and 16-byte store
        for (uint32_t w = sp.w0 + 4 * threadIdx.x; w < sp.w1; w += 4 * blockDim.x) {
            uint32_t a[5];
            load5(a, true, i0 + (w - sp.w0), sh, srcWord);
            store4(a, sh, w);
        }
    };
    // The spans of a compressed packet at once — shift-off bytes (24- / 32-bit), U bits, V bits (stereo): the loads of
    // ALL are issued before anything is stored, so a wave pays one memory round trip for its share of the packet.
    // Shift-off bytes (codec/ALACEncoder.cu:489-503) = the low SHB bytes of every channel sample, MSB first:
    // 24-bit: one byte per 3-byte sample, four samples (three PCM dwords) per output word, read as three 16-byte loads
    // per group of four words; 32-bit: two bytes per sample, two samples (two PCM dwords) per output word.
    typedef uint32_t U4 __attribute__((ext_vector_type(4), aligned(4)));
    auto shift_word24 = [](uint32_t d0, uint32_t d1, uint32_t d2) {
        return (d0 << 24) | ((d0 >> 24) << 16) | (((d1 >> 16) & 0xffu) << 8) | ((d2 >> 8) & 0xffu);
    };
    auto shift_word_slow = [&](uint32_t i) -> uint32_t {  // a word at the ragged end of the samples
        const uint32_t fields = N * CH;
        if constexpr (DEPTH == 24) {
            uint32_t v = 0;
            for (uint32_t q = 0; q < 4; q++)
                v = (v << 8) | (4 * i + q < fields ? ((uint32_t)load_sample<24>(pk, 4 * i + q) & 0xffu) : 0u);
            return v;
        } else {
            const uint32_t *pw = (const uint32_t *)pk;
            const uint32_t t = 2 * i;
            const uint32_t a = t < fields ? pw[t] : 0u, b = t + 1 < fields ? pw[t + 1] : 0u;
            return (a << 16) | (b & 0xffffu);
        }
    };
    // source words i .. i + 3 are complete (the output group lies inside the segment); i + 4 may be ragged
    auto load5_shift = [&](uint32_t a[5], bool have, uint32_t i, uint32_t sh) {
        const uint32_t *pw = (const uint32_t *)pk;  // packets are dword aligned for these depths (frame * 6 or 8 bytes)
        const uint32_t fields = N * CH;
#pragma unroll
        for (int q = 0; q < 5; q++) a[q] = 0;
        if (!have) return;
        if constexpr (DEPTH == 24) {
            const char *at = (const char *)pw + i * 12u;
            const U4 x = *(const U4 *)at, y = *(const U4 *)(at + 16), z = *(const U4 *)(at + 32);
            a[0] = shift_word24(x.x, x.y, x.z);
            a[1] = shift_word24(x.w, y.x, y.y);
            a[2] = shift_word24(y.z, y.w, z.x);
            a[3] = shift_word24(z.y, z.z, z.w);
            if (sh) {
                if (4 * (i + 4) + 4 <= fields) {
                    const uint32_t d0 = *(const uint32_t *)(at + 48), d1 = *(const uint32_t *)(at + 52), d2 = *(const uint32_t *)(at + 56);
                    a[4] = shift_word24(d0, d1, d2);
                } else {
                    a[4] = shift_word_slow(i + 4);
                }
            }
        } else {
            const char *at = (const char *)pw + i * 8u;
            const U4 x = *(const U4 *)at, y = *(const U4 *)(at + 16);
            a[0] = (x.x << 16) | (x.y & 0xffffu);
            a[1] = (x.z << 16) | (x.w & 0xffffu);
            a[2] = (y.x << 16) | (y.y & 0xffffu);
            a[3] = (y.z << 16) | (y.w & 0xffffu);
            if (sh) {
                if (2 * (i + 4) + 2 <= fields) {
                    const uint32_t d0 = *(const uint32_t *)(at + 32), d1 = *(const uint32_t *)(at + 36);
                    a[4] = (d0 << 16) | (d1 & 0xffffu);
                } else {
                    a[4] = shift_word_slow(i + 4);
                }
            }
        }
    };
    auto copy_bulk = [&](const Span &ss, const Span &sa, const Span &sb) {
        constexpr bool SHIFT = SHB != 0 && (DEPTH == 24 || DEPTH == 32);
        const uint32_t fs = first_bit(ss, offShift), is0 = fs >> 5, shS = fs & 31;
        const uint32_t fa = first_bit(sa, offU), ia0 = fa >> 5, shA = fa & 31;
        const uint32_t fb = first_bit(sb, offV), ib0 = fb >> 5, shB = fb & 31;
        const uint32_t ns = SHIFT ? (ss.w1 - ss.w0) / 4 : 0u, na = (sa.w1 - sa.w0) / 4, nb2 = CH == 2 ? (sb.w1 - sb.w0) / 4 : 0u;
        for (uint32_t g = threadIdx.x; g < max(ns, max(na, nb2)); g += blockDim.x) {
            const bool hs = g < ns, ha = g < na, hb = g < nb2;
            uint32_t x[5], a[5], b[5];
            if constexpr (SHIFT) load5_shift(x, hs, is0 + 4 * g, shS);
            load5(a, ha, ia0 + 4 * g, shA, wU);
            if constexpr (CH == 2) load5(b, hb, ib0 + 4 * g, shB, wV);
            if constexpr (SHIFT)
                if (hs) store4(x, shS, ss.w0 + 4 * g);
            if (ha) store4(a, shA, sa.w0 + 4 * g);
            if constexpr (CH == 2)
                if (hb) store4(b, shB, sb.w0 + 4 * g);
        }
    };
    Span spU = {0, 0}, spV = {0, 0}, spR = {0, 0}, spS = {0, 0};
    if (!rec.escape) {
        if constexpr (SHB != 0 && (DEPTH == 24 || DEPTH == 32)) spS = span_of(offShift, lenShift);
        spU = span_of(offU, lenU);
        if constexpr (CH == 2) spV = span_of(offV, lenV);
    } else {
        spR = span_of(offRaw, lenRaw);
    }
    // Seam words = the words no span covers: in front of the first span (header), between spans, the tail with ID_END.
    // Lane k of the last wave owns seam word k; its operands from the U and V bit strings are requested BEFORE the
    // wave takes part in the bulk copy and combined after it, so the wave pays one memory round trip per packet, too.
    const Span s1 = rec.escape ? spR : spS, s2 = spU, s3 = spV;  // in stream order, empty ones have w1 == w0
    const uint32_t nSeam = nwords - (s1.w1 - s1.w0) - (s2.w1 - s2.w0) - (s3.w1 - s3.w0);
    auto seam_word = [&](uint32_t k) {
        uint32_t aw = k;
        if (aw >= s1.w0) aw += s1.w1 - s1.w0;
        if (aw >= s2.w0) aw += s2.w1 - s2.w0;
        if (aw >= s3.w0) aw += s3.w1 - s3.w0;
        return aw;
    };
    // 32 bits of a word string from bit s (may be negative / beyond the end): both candidate words, clamped, no branch
    struct Pair {
        uint32_t a, b;
    };
    auto fetch_pair = [&](const uint32_t *w, uint64_t len, int64_t s) {
        const uint32_t nw = (uint32_t)((len + 31) >> 5);
        const uint32_t i = s > 0 ? (uint32_t)(s >> 5) : 0u;
        const uint32_t last = nw ? nw - 1 : 0u;
        Pair pr;
        pr.a = w[min(i, last)];
        pr.b = w[min(i + 1, last)];
        return pr;
    };
    auto combine_pair = [&](const Pair &pr, uint64_t len, int64_t s) -> uint32_t {
        if (s >= (int64_t)len || s <= -32) return 0u;
        const uint32_t s0 = s > 0 ? (uint32_t)s : 0u;
        const uint32_t i = s0 >> 5, sh = s0 & 31;
        const uint32_t nw = (uint32_t)((len + 31) >> 5);
        const uint32_t b = (i + 1 < nw) ? pr.b : 0u;
        uint32_t v = sh ? ((pr.a << sh) | (b >> (32 - sh))) : pr.a;
        const uint32_t rem = (uint32_t)len - s0;
        if (rem < 32) v &= ~0u << (32 - rem);
        return s < 0 ? v >> (uint32_t)(-s) : v;
    };
    auto seam_store = [&](uint32_t aw, uint32_t v) {
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
    };
    // 32 bits of the shift-off byte string from bit s0 (0 <= s0 < lenShift): both source words requested at once
    auto shift_fetch32 = [&](uint32_t s0) -> uint32_t {
        if constexpr (SHB == 0) {
            return 0u;
        } else if constexpr (DEPTH != 24 && DEPTH != 32) {
            return fields_fetch32<SHB * 8>(sampleField, N * CH, s0);
        } else {
            const uint32_t *pw = (const uint32_t *)pk;
            const uint32_t fields = N * CH, perWord = DEPTH == 24 ? 4u : 2u, dwPerWord = DEPTH == 24 ? 3u : 2u;
            const uint32_t i = s0 >> 5, sh = s0 & 31;
            const uint32_t nFull = fields / perWord;  // complete source words
            const bool fullA = i < nFull, fullB = i + 1 < nFull;
            // clamped, unconditional: the ragged last word (if any) takes the slow path afterwards
            const uint32_t ia = fullA ? i : 0u, ib = fullB ? i + 1 : 0u;
            uint32_t da[3], db[3];
#pragma unroll
            for (uint32_t q = 0; q < dwPerWord; q++) da[q] = nFull ? pw[ia * dwPerWord + q] : 0u;
#pragma unroll
            for (uint32_t q = 0; q < dwPerWord; q++) db[q] = nFull ? pw[ib * dwPerWord + q] : 0u;
            uint32_t a, b;
            if constexpr (DEPTH == 24) {
                a = shift_word24(da[0], da[1], da[2]);
                b = shift_word24(db[0], db[1], db[2]);
            } else {
                a = (da[0] << 16) | (da[1] & 0xffffu);
                b = (db[0] << 16) | (db[1] & 0xffffu);
            }
            if (!fullA) a = shift_word_slow(i);
            if (!fullB) b = shift_word_slow(i + 1);  // zero beyond the samples
            uint32_t v = sh ? ((a << sh) | (b >> (32 - sh))) : a;
            const uint32_t rem = (uint32_t)lenShift - s0;
            if (rem < 32) v &= ~0u << (32 - rem);
            return v;
        }
    };
    // everything of a seam word except the U / V operands
    auto seam_rest = [&](int64_t bp) {
        uint32_t v = take([&](uint64_t s0) { return words_fetch32(hdr, (uint32_t)lenHdr, (uint32_t)s0); }, lenHdr, bp);
        if (!rec.escape) {
            if constexpr (SHB != 0)
                v |= take([&](uint64_t s0) { return shift_fetch32((uint32_t)s0); }, lenShift, bp - offShift);
        } else {
            v |= take([&](uint64_t s0) { return fields_fetch32<DEPTH>(sampleField, N * CH, s0); }, lenRaw, bp - offRaw);
        }
        v |= take([](uint64_t s0) { return 0xE0000000u << (uint32_t)s0; }, 3, bp - offEnd);
        return v;
    };
    Pair seamU = {0, 0}, seamV = {0, 0};
    const uint32_t seamAw = seam_word(lane);
    const int64_t seamBp = (int64_t)seamAw * 32 - (int64_t)mis * 8;
    if (seamWave) {
        // ---- header (Encode() :989-991 / :1011-1012 element tag, then the element header): one lane per field, OR-ed
        // into zeroed LDS words.  lane 0: tag 3 | 0 4 | 0 12 | flags 4 (:467 / :762 / :921); lane 1: numSamples of a
        // partial frame; lane 2: mixBits | mixRes; per channel c: lane 3 + 9c = (mode|denShift, pbFactor|num), lanes
        // 4 + 9c + k = coefficient k (picked out of the record's registers by a select chain: no load).
        if (lane < 16) hdr[lane] = 0;
        uint32_t fpos = 0, fn = 0, fv = 0;
        const uint32_t base = 23 + (partial ? 32u : 0u);
        if (lane == 0) {
            fn = 23;
            fv = ((CH == 2 ? 1u : 0u) << 20) | (rec.escape ? ((partial << 3) | 1u) : ((partial << 3) | (SHB << 1)));
        } else if (lane == 1) {
            fpos = 23, fn = partial ? 32u : 0u, fv = N;
        } else if (!rec.escape) {
            if (lane == 2) {
                fpos = base, fn = 16, fv = CH == 2 ? (((uint32_t)kMixBits << 8) | (rec.mixRes & 0xffu)) : 0u;
            } else if (lane < 3 + 9 * CH) {
                const uint32_t c = (lane - 3) / 9, k = (lane - 3) - 9 * c;
                const uint32_t num = c ? rec.c[CH - 1].num : rec.c[0].num;
                const uint32_t cstart = base + 16 + (c ? 16 + 16 * (uint32_t)rec.c[0].num : 0u);
                if (k == 0) {
                    fpos = cstart, fn = 16, fv = (((0u << 4) | kDenShift) << 8) | ((4u << 5) | num);
                } else if (k - 1 < num) {
                    uint32_t coef = 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint32_t cq = (uint16_t)(c ? rec.c[CH - 1].coefs[q] : rec.c[0].coefs[q]);
                        coef = (k - 1 == (uint32_t)q) ? cq : coef;
                    }
                    fpos = cstart + 16 * k, fn = 16, fv = coef;
                }
            }
        }
        if (fn) {
            const uint32_t i = fpos >> 5, o = fpos & 31;
            const uint64_t x = ((uint64_t)fv << (64 - fn)) >> o;  // field left-aligned at bit o of a 64-bit window
            atomicOr(&hdr[i], (uint32_t)(x >> 32));
            if (o + fn > 32) atomicOr(&hdr[i + 1], (uint32_t)x);
        }
        if (!rec.escape && lane < nSeam) {
            seamU = fetch_pair(wU, lenU, seamBp - offU);
            if constexpr (CH == 2) seamV = fetch_pair(wV, lenV, seamBp - offV);
        }
    }
    if (!rec.escape) {
        copy_bulk(spS, spU, spV);
    } else if constexpr (DEPTH == 16) {
        // raw 16-bit samples, MSB first: two little-endian samples per PCM word, swapped into place by a rotate
        const uint32_t *pw = (const uint32_t *)pk;  // 16-byte aligned packet (checked by the C entry point)
        copy_span(spR, offRaw, [&](uint32_t i) {
            const uint32_t x = pw[i];
            return (x << 16) | (x >> 16);
        });
    } else if constexpr (DEPTH == 32) {
        // raw 32-bit samples, MSB first: the word IS the little-endian sample
        const uint32_t *pw = (const uint32_t *)pk;
        copy_span(spR, offRaw, [&](uint32_t i) { return pw[i]; });
    } else if constexpr (DEPTH == 24) {
        // raw 24-bit samples, MSB first: four samples (PCM bytes B0..B11 in three dwords) make three words
        // B2 B1 B0 B5 | B4 B3 B8 B7 | B6 B11 B10 B9.  A lane's five source words i .. i + 4 lie in three such triples:
        // nine dwords loaded unconditionally (two 16-byte loads + one), all nine words formed, five picked by i mod 3.
        const uint32_t *pw = (const uint32_t *)pk;
        const uint32_t ndw = (N * CH * 3 + 3) / 4;  // dwords that hold samples (the tail of the last one is padding)
        const uint32_t fb = first_bit(spR, offRaw), i0 = fb >> 5, sh = fb & 31;
        for (uint32_t w = spR.w0 + 4 * threadIdx.x; w < spR.w1; w += 4 * blockDim.x) {
            const uint32_t i = i0 + (w - spR.w0);
            const uint32_t g = i / 3, r = i - 3 * g, base = 3 * g;
            uint32_t d[9];
            if (base + 9 <= ndw) {
                const U4 x = *(const U4 *)(pw + base), y = *(const U4 *)(pw + base + 4);
                d[0] = x.x, d[1] = x.y, d[2] = x.z, d[3] = x.w, d[4] = y.x, d[5] = y.y, d[6] = y.z, d[7] = y.w;
                d[8] = pw[base + 8];
            } else {
                // the end of the samples: dwords past it feed only words (or low bits of the fifth word) nobody uses
#pragma unroll
                for (uint32_t q = 0; q < 9; q++) d[q] = pw[min(base + q, ndw - 1)];
            }
            uint32_t W[9];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const uint32_t d0 = d[3 * t], d1 = d[3 * t + 1], d2 = d[3 * t + 2];
                W[3 * t] = (d0 << 8) | ((d1 >> 8) & 0xffu);
                W[3 * t + 1] = (d1 << 24) | ((d0 >> 24) << 16) | ((d2 & 0xffu) << 8) | (d1 >> 24);
                W[3 * t + 2] = (((d1 >> 16) & 0xffu) << 24) | ((d2 >> 24) << 16) | (((d2 >> 16) & 0xffu) << 8) | ((d2 >> 8) & 0xffu);
            }
            uint32_t a[5];
#pragma unroll
            for (int q = 0; q < 5; q++) a[q] = r == 0 ? W[q] : (r == 1 ? W[q + 1] : W[q + 2]);
            store4(a, sh, w);
        }
    } else {
        // raw 20-bit samples, MSB first: eight samples (24 PCM bytes, six dwords) make five words.  A lane's five source
        // words lie in two such periods: twelve dwords, sixteen samples, ten words, five picked by i mod 5.
        const uint32_t *pw = (const uint32_t *)pk;
        const uint32_t ndw = (N * CH * 3 + 3) / 4;
        const uint32_t fb = first_bit(spR, offRaw), i0 = fb >> 5, sh = fb & 31;
        for (uint32_t w = spR.w0 + 4 * threadIdx.x; w < spR.w1; w += 4 * blockDim.x) {
            const uint32_t i = i0 + (w - spR.w0);
            const uint32_t g = i / 5, r = i - 5 * g, base = 6 * g;
            uint32_t d[12];
            if (base + 12 <= ndw) {
                const U4 x = *(const U4 *)(pw + base), y = *(const U4 *)(pw + base + 4), z = *(const U4 *)(pw + base + 8);
                d[0] = x.x, d[1] = x.y, d[2] = x.z, d[3] = x.w, d[4] = y.x, d[5] = y.y, d[6] = y.z, d[7] = y.w;
                d[8] = z.x, d[9] = z.y, d[10] = z.z, d[11] = z.w;
            } else {
#pragma unroll
                for (uint32_t q = 0; q < 12; q++) d[q] = pw[min(base + q, ndw - 1)];
            }
            uint32_t W[10];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                uint32_t sm[8];  // the 20-bit fields of eight samples (top 20 bits of each 3-byte little-endian sample)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t d0 = d[6 * t + 3 * h], d1 = d[6 * t + 3 * h + 1], d2 = d[6 * t + 3 * h + 2];
                    sm[4 * h] = (d0 & 0xffffffu) >> 4;
                    sm[4 * h + 1] = (((d0 >> 24) | (d1 << 8)) & 0xffffffu) >> 4;
                    sm[4 * h + 2] = (((d1 >> 16) | (d2 << 16)) & 0xffffffu) >> 4;
                    sm[4 * h + 3] = d2 >> 12;
                }
                W[5 * t] = (sm[0] << 12) | (sm[1] >> 8);
                W[5 * t + 1] = (sm[1] << 24) | (sm[2] << 4) | (sm[3] >> 16);
                W[5 * t + 2] = (sm[3] << 16) | (sm[4] >> 4);
                W[5 * t + 3] = (sm[4] << 28) | (sm[5] << 8) | (sm[6] >> 12);
                W[5 * t + 4] = (sm[6] << 20) | sm[7];
            }
            uint32_t a[5];
#pragma unroll
            for (int q = 0; q < 5; q++)
                a[q] = r == 0 ? W[q] : (r == 1 ? W[q + 1] : (r == 2 ? W[q + 2] : (r == 3 ? W[q + 3] : W[q + 4])));
            store4(a, sh, w);
        }
    }

    if (seamWave) {
        if (lane < nSeam) {
            uint32_t v = seam_rest(seamBp);
            if (!rec.escape) {
                v |= combine_pair(seamU, lenU, seamBp - offU);
                if constexpr (CH == 2) v |= combine_pair(seamV, lenV, seamBp - offV);
            }
            seam_store(seamAw, v);
        }
        for (uint32_t k = lane + 64; k < nSeam; k += 64) {  // more than 64 seam words: tiny packets of odd shapes only
            const uint32_t aw = seam_word(k);
            const int64_t bp = (int64_t)aw * 32 - (int64_t)mis * 8;
            uint32_t v = seam_rest(bp);
            if (!rec.escape) {
                v |= combine_pair(fetch_pair(wU, lenU, bp - offU), lenU, bp - offU);
                if constexpr (CH == 2) v |= combine_pair(fetch_pair(wV, lenV, bp - offV), lenV, bp - offV);
            }
            seam_store(aw, v);
        }
    }
    }  // packets of this workgroup
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

template <int DEPTH>
static void launch_scan_pack_depth(uint32_t channels, uint32_t *packetBytes, const PackArgs &pa, uint32_t numPackets,
                                   hipStream_t st, hipEvent_t *ev, bool recordScan = true)
{
    if (ev && recordScan) (void)hipEventRecord(ev[kStageScan], st);
    hipLaunchKernelGGL(k_scan_sizes, dim3(numPackets / 1024 + 1), dim3(1024), 0, st, (const uint32_t *)packetBytes, (uint64_t *)pa.offsets,
                       numPackets, pa.segBad);
    if (ev) (void)hipEventRecord(ev[kStagePack], st);
    // One workgroup of 128 threads per packet (ALAC_HIP_PACK_TPB / ALAC_HIP_PACK_WGS override; with fewer workgroups than
    // packets each walks several).  Measured at 10 000 16-bit packets / 125 000 (tools/dispatch_rate_microbench.hip for the
    // plain-copy floor): 64 threads 0.064 / 0.66 ms, 128: 0.064 / 0.66, 256: 0.067 / 0.74; 2048 persistent workgroups
    // 0.079 / 0.93; a plain 6.8 KB-per-workgroup copy of the same bytes 0.017 / 0.29.  Workgroup dispatch is not the limit
    // (125 000 empty workgroups launch in 0.03 ms).  What round 2 removed: a workgroup-wide barrier behind a header built by
    // one lane, a second pass over all words for the seams, up to four dependent loads in front of / behind every span, and
    // — the big one for 20- / 24-bit — sample-by-sample loops with a dependent load per step (an escape packet took 180 us).
    static const uint32_t tpb = getenv("ALAC_HIP_PACK_TPB") ? (uint32_t)atoi(getenv("ALAC_HIP_PACK_TPB")) : 128u;
    static const uint32_t wgs = getenv("ALAC_HIP_PACK_WGS") ? (uint32_t)atoi(getenv("ALAC_HIP_PACK_WGS")) : (1u << 20);
    const uint32_t grid = numPackets < wgs ? numPackets : wgs;
    if (channels == 2)
        hipLaunchKernelGGL((k_pack<DEPTH, 2>), dim3(grid), dim3(tpb), 0, st, pa, numPackets);
    else
        hipLaunchKernelGGL((k_pack<DEPTH, 1>), dim3(grid), dim3(tpb), 0, st, pa, numPackets);
    if (ev) (void)hipEventRecord(ev[kNumStages], st);
}

void launch_scan_sizes(const uint32_t *sizes, uint64_t *offsets, uint32_t n, hipStream_t st, const uint32_t *segBad)
{
    hipLaunchKernelGGL(k_scan_sizes, dim3(n / 1024 + 1), dim3(1024), 0, st, sizes, offsets, n, segBad);
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
