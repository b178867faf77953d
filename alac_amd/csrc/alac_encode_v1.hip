// alac_encode_v1.hip — the tap-parallel encode pipeline.
//
// Per "packet position" of the segments (position 0 only when every packet is its own segment):
//   k_lms_search1   LPC+mix: the five mixRes passes over N/8 samples walking row 7 (stereo only)
//   k_gol_count     adaptive-Golomb bit counts of those passes, one lane per (chain, pass) stream
//   k_decide1       best mixRes per packet
//   k_lms_search2   LPC+mix: the 8 converge passes of rows 3 and 7 (mono: 7 + 1 passes)
//   k_gol_count     bit counts (first part from the last converge pass, tail from the mixRes = 4 pass:
//                   the stale-predictor-tail quirk of SURVEY.md §3.2)
//   k_decide2       numU / numV, escape estimate, header coefficients
//   k_lms_final     LPC+mix: the final pass over N samples with the chosen row
//   k_gol_final     adaptive-Golomb coder writing the per-channel bit strings, one lane per chain
// then once: k_finalize (sizes + the "too big -> escape" rule), k_scan_sizes, k_pack (alac_encode.hip).
//
// The LPC+mix kernels are the tap-parallel form of alac_lms.hpp: one wave = 8 chains x 8 taps,
// inputs staged in LDS tile by tile, residual tiles flushed to HBM in [sample][stream] order so that
// the lane-per-stream Golomb kernels read them coalesced.
//
// Reference control flow: codec/ALACEncoder.cu:290-558 (EncodeStereo), :812-963 (EncodeMono).
#include "alac_dev.hpp"
#include "alac_kernels.hpp"
#include "alac_lms.hpp"
#include "alac_golomb.hpp"

namespace alacdev {

constexpr int kTile = 128;        // predictor steps per LDS tile
constexpr int kHist = 9;          // history kept in front of a tile: in[j-9] is "top" for 8 taps
constexpr int kXsStride = 168;    // dwords per xs row: >= kHist + kTile, == 8 (mod 32): 4 groups hit disjoint banks
constexpr int kResStride = 136;   // dwords per residual row, == 8 (mod 32)

struct SegView {
    const uint8_t *pcm;
    const uint32_t *numSamples;
    const uint32_t *segFirst;
    uint32_t numSegments, frameSize, pos;
};

__device__ __forceinline__ bool seg_packet(const SegView &S, uint32_t seg, uint32_t &p, uint32_t &N)
{
    p = 0;
    N = 0;
    if (seg >= S.numSegments) return false;
    const uint32_t p0 = S.segFirst ? S.segFirst[seg] : seg;
    const uint32_t p1 = S.segFirst ? S.segFirst[seg + 1] : seg + 1;
    p = p0 + S.pos;
    if (p >= p1) return false;
    N = S.numSamples ? S.numSamples[p] : S.frameSize;
    N = N < S.frameSize ? N : S.frameSize;
    return true;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// ---- LDS tile staging ---------------------------------------------------------------------------
// Fill xs row `row` with x[j0 - kHist .. j0 + kTile) of one chain input: stereo u or v for `mixres`
// (codec/matrix_enc.cu:72-99 and the 20/24/32-bit forms), or the mono sample.  Out-of-range -> 0.
template <int DEPTH, int CH>
__device__ __forceinline__ void stage_rows(int32_t *xs, int rowU, int rowV, const uint8_t *pk, uint32_t N, int mixres,
                                           int j0, int lane)
{
    constexpr int SH = 8 * (int)bytes_shifted(DEPTH);
    for (int i = lane; i < kHist + kTile; i += 64) {
        const int j = j0 - kHist + i;
        int32_t u = 0, v = 0;
        if (j >= 0 && j < (int)N) {
            if constexpr (CH == 2) {
                int32_t l, r;
                load_lr<DEPTH>(pk, (uint32_t)j, l, r);
                u = mix_sample(mixres, 0, l, r);
                v = mix_sample(mixres, 1, l, r);
            } else {
                u = load_sample<DEPTH>(pk, (uint32_t)j) >> SH;
            }
        }
        xs[rowU * kXsStride + i] = u;
        if constexpr (CH == 2) xs[rowV * kXsStride + i] = v;
    }
}

// ---- one tile of predictor steps for the 8 chains of the wave --------------------------------------
// p[j] = in[j] - in[j-1-na] is the same for the 8 taps of a chain: computed once per sample here (8 lanes of
// the slot x 16 rounds) instead of once per lane per step inside the recurrence loop.
__device__ __forceinline__ void stage_p(const int32_t *xsRow, int32_t *pRow, int na, int k)
{
#pragma unroll
    for (int i = 0; i < (kTile + 8) / 8; i++) {
        const int idx = i * 8 + k;  // sample j0 + idx
        pRow[idx] = xsRow[kHist + idx] - xsRow[kHist + idx - 1 - na];
    }
}

// The LDS operands of a step (in[j-1-k], top, p) depend on nothing the recurrence produces, so the
// operands of block i+1 (8 steps) are fetched into registers while block i computes.
struct StepOps {
    int32_t xk[8], tp[8], p[8];
};

__device__ __forceinline__ void load_ops(StepOps &o, const int32_t *px, const int32_t *pt, const int32_t *pp, int jb)
{
#pragma unroll
    for (int s = 0; s < 8; s++) {
        o.xk[s] = px[jb + s];
        o.tp[s] = pt[jb + s];
        o.p[s] = pp[jb + s];
    }
}

template <bool WIDE, bool MASKED>
__device__ __forceinline__ void run_block(int32_t &a, const StepOps &o, const LmsLane &L, int jb, int32_t *resAt,
                                          uint32_t chanbits)
{
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int j = jb + s;
        const int32_t liveMask = MASKED ? (((j >= L.jlo) & (j < L.jhi)) ? -1 : 0) : -1;
        // the residual is identical in the 8 lanes of the group: all of them store it (same address, no VALU)
        resAt[s] = lms8_step<WIDE, MASKED>(a, o.xk[s], o.tp[s], o.p[s], liveMask, L, chanbits);
    }
}

template <bool WIDE>
__device__ __forceinline__ void run_tile(int32_t &a, const int32_t *xsRow, const int32_t *pRow, int32_t *resRow,
                                         const LmsLane &L, int j0, int jEnd, uint32_t chanbits)
{
    // per-lane LDS cursors: x[j] lives at xsRow[kHist + j - j0]; rows are long enough for the one-block
    // over-read of the prefetch
    const int kk = L.k < L.na ? L.k : L.na;  // inert taps read "top" so their b is 0
    const int32_t *px = xsRow + kHist - 1 - kk - j0;
    const int32_t *pt = xsRow + kHist - 1 - L.na - j0;
    const int32_t *pp = pRow - j0;
    StepOps cur, nxt;
    load_ops(cur, px, pt, pp, j0);
    for (int jb = j0; jb < jEnd; jb += 8) {
        load_ops(nxt, px, pt, pp, jb + 8);
        // wave-uniform: is every lane's chain live for the whole block?
        const bool allLive = __all((jb >= L.jlo) & (jb + 8 <= L.jhi));
        if (allLive)
            run_block<WIDE, false>(a, cur, L, jb, resRow + (jb - j0), chanbits);
        else
            run_block<WIDE, true>(a, cur, L, jb, resRow + (jb - j0), chanbits);
        cur = nxt;
    }
}

// warm-up positions of pc_block (dp_enc.c:90, :108-112): pc[0] = in[0], pc[j] = sext(in[j] - in[j-1]), j <= na
__device__ __forceinline__ void warmup_fix(const int32_t *xsRow, int32_t *resRow, const LmsLane &L, uint32_t chanshift)
{
    for (int pos = L.k; pos <= L.na; pos += 8) {
        const int32_t x = xsRow[kHist + pos];
        resRow[pos] = pos == 0 ? x : sext(x - xsRow[kHist + pos - 1], chanshift);
    }
}

// residual tile -> HBM, [sample][stream] layout
__device__ __forceinline__ void flush_tile(const int32_t *resRow, int32_t *dst, uint64_t streamStride, uint32_t stream,
                                           int j0, uint32_t P, int k)
{
    int32_t v[kTile / 8];
#pragma unroll
    for (int i = 0; i < kTile / 8; i++) v[i] = resRow[i * 8 + k];
#pragma unroll
    for (int i = 0; i < kTile / 8; i++) {
        const uint32_t j = (uint32_t)(j0 + i * 8 + k);
        if (j < P) dst[(uint64_t)j * streamStride + stream] = v[i];
    }
}

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
    uint32_t *packetBytes;
};

// ================================================================================================
// k_lms_search1 — stereo mixRes search passes (codec/ALACEncoder.cu:353-379)
// ================================================================================================
template <int DEPTH>
__global__ __launch_bounds__(64) void k_lms_search1(V1Args A)
{
    __shared__ int32_t xs[8 * kXsStride];
    __shared__ int32_t ps[8 * kResStride];
    __shared__ int32_t res[8 * kResStride];
    const int lane = threadIdx.x;
    const int g = lane >> 3, q = g >> 1, c = g & 1;
    const uint32_t seg = blockIdx.x * 4u + q;
    uint32_t p, N;
    const bool active = seg_packet(A.S, seg, p, N);
    const uint32_t n8 = N / 8;
    constexpr uint32_t chanBits = DEPTH - 8 * bytes_shifted(DEPTH) + 1;
    constexpr bool WIDE = chanBits > 17;
    const uint32_t frameBytes = A.S.frameSize * 2u * bytes_per_sample(DEPTH);

    LmsLane L = make_lane(lane, 8, (int)n8);
    int16_t *row7 = A.state + (uint64_t)seg * 64 + c * 32 + 16;
    int32_t a = active ? (int32_t)row7[L.k] : 0;
    if (!active) L.jhi = 0;
    const uint32_t maxn8 = wave_max(active ? n8 : 0);
    const uint32_t chain = seg * 2 + c;

    for (int r = 0; r <= kMaxRes; r++) {
        for (int j0 = 0; j0 < (int)maxn8; j0 += kTile) {
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const uint32_t pq = (uint32_t)__builtin_amdgcn_readlane((int)p, qq * 16);
                const uint32_t Nq = (uint32_t)__builtin_amdgcn_readlane((int)N, qq * 16);
                stage_rows<DEPTH, 2>(xs, 2 * qq, 2 * qq + 1, A.S.pcm + (uint64_t)pq * frameBytes, Nq, r, j0, lane);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            stage_p(xs + g * kXsStride, ps + g * kResStride, L.na, L.k);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const int jEnd = min(j0 + kTile, (int)((maxn8 + 7) & ~7u));
            run_tile<WIDE>(a, xs + g * kXsStride, ps + g * kResStride, res + g * kResStride, L, j0, jEnd, chanBits);
            if (j0 == 0) warmup_fix(xs + g * kXsStride, res + g * kResStride, L, 32 - chanBits);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (active)
                flush_tile(res + g * kResStride, A.resA, 5ull * A.chainsPad, (uint32_t)r * A.chainsPad + chain, j0, n8, L.k);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    }
    if (active) row7[L.k] = (int16_t)a;
}

// ================================================================================================
// k_lms_search2 — converge passes for numUV = 4 and 8 (codec/ALACEncoder.cu:420-431; mono :881-893)
// ================================================================================================
template <int DEPTH, int CH>
__global__ __launch_bounds__(64) void k_lms_search2(V1Args A)
{
    __shared__ int32_t xs[4 * kXsStride];
    __shared__ int32_t ps[8 * kResStride];
    __shared__ int32_t res[8 * kResStride];
    const int lane = threadIdx.x;
    const int g = lane >> 3;
    const int rs = g & 1;                          // 0: row 3 (4 taps), 1: row 7 (8 taps)
    const int xrow = g >> 1;                       // LDS input row shared by the two rows of a chain
    const int q = CH == 2 ? (g >> 2) : (g >> 1);   // segment slot inside the wave
    const int c = CH == 2 ? ((g >> 1) & 1) : 0;
    constexpr int SEGS = CH == 2 ? 2 : 4;
    const uint32_t seg = blockIdx.x * SEGS + q;
    uint32_t p, N;
    const bool active = seg_packet(A.S, seg, p, N);
    const uint32_t n8 = N / 8, n32 = N / 32;
    constexpr uint32_t chanBits = DEPTH - 8 * bytes_shifted(DEPTH) + (CH == 2 ? 1 : 0);
    constexpr bool WIDE = chanBits > 17;
    const uint32_t frameBytes = A.S.frameSize * CH * bytes_per_sample(DEPTH);
    const int na = rs ? 8 : 4;

    LmsLane L = make_lane(lane, na, (int)n32);
    int16_t *row = A.state + (uint64_t)seg * 64 + c * 32 + rs * 16;
    int32_t a = (active && L.k < na) ? (int32_t)row[L.k] : 0;
    const int best = (CH == 2 && active) ? (int)A.recs[p].mixRes : 0;
    const uint32_t chain = seg * CH + c;

    for (int pass = 0; pass < 8; pass++) {
        const bool last = pass == 7;
        // stereo: every pass runs N/32 samples; mono: the last one runs N/8 (:893)
        const uint32_t num = (CH == 1 && last) ? n8 : n32;
        uint32_t P = num > (uint32_t)(na + 1) ? num : (uint32_t)(na + 1);  // positions pc_block writes
        P = P < n8 ? P : n8;                                               // ... that dyn_comp will read
        L.jhi = active ? (int)num : 0;
        const uint32_t runTo = wave_max(active ? (last ? (P > num ? P : num) : num) : 0);
        if (!last && runTo <= 5) continue;  // nothing adapts (num <= na + 1 for every chain)
        for (int j0 = 0; j0 < (int)runTo; j0 += kTile) {
#pragma unroll
            for (int xr = 0; xr < 4; xr++) {
                constexpr int lanesPerX = 16;  // two slots (rows 3 and 7) share an input row
                const uint32_t pq = (uint32_t)__builtin_amdgcn_readlane((int)p, xr * lanesPerX);
                const uint32_t Nq = (uint32_t)__builtin_amdgcn_readlane((int)N, xr * lanesPerX);
                const int bq = __builtin_amdgcn_readlane(best, xr * lanesPerX);
                const int cq = CH == 2 ? (xr & 1) : 0;
                const uint8_t *pk = A.S.pcm + (uint64_t)pq * frameBytes;
                if constexpr (CH == 2) {
                    // rows xr = 2*q' + c': stage only the channel this row carries
                    constexpr int SH = 8 * (int)bytes_shifted(DEPTH);
                    (void)SH;
                    for (int i = lane; i < kHist + kTile; i += 64) {
                        const int j = j0 - kHist + i;
                        int32_t x = 0;
                        if (j >= 0 && j < (int)Nq) {
                            int32_t l, r;
                            load_lr<DEPTH>(pk, (uint32_t)j, l, r);
                            x = mix_sample(bq, cq, l, r);
                        }
                        xs[xr * kXsStride + i] = x;
                    }
                } else {
                    stage_rows<DEPTH, 1>(xs, xr, xr, pk, Nq, 0, j0, lane);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            stage_p(xs + xrow * kXsStride, ps + g * kResStride, L.na, L.k);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            const int jEnd = min(j0 + kTile, (int)((runTo + 7) & ~7u));
            run_tile<WIDE>(a, xs + xrow * kXsStride, ps + g * kResStride, res + g * kResStride, L, j0, jEnd, chanBits);
            if (last) {
                if (j0 == 0) warmup_fix(xs + xrow * kXsStride, res + g * kResStride, L, 32 - chanBits);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                if (active)
                    flush_tile(res + g * kResStride, A.resB, 2ull * A.chainsPad, (uint32_t)rs * A.chainsPad + chain, j0, P, L.k);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    }
    if (active && L.k < na) row[L.k] = (int16_t)a;
}

// ================================================================================================
// k_lms_final — final predictor pass with the chosen row (codec/ALACEncoder.cu:505-532, :941)
// ================================================================================================
template <int DEPTH, int CH>
__global__ __launch_bounds__(64) void k_lms_final(V1Args A)
{
    __shared__ int32_t xs[8 * kXsStride];
    __shared__ int32_t ps[8 * kResStride];
    __shared__ int32_t res[8 * kResStride];
    const int lane = threadIdx.x;
    const int g = lane >> 3;
    const int q = CH == 2 ? (g >> 1) : g;
    const int c = CH == 2 ? (g & 1) : 0;
    constexpr int SEGS = CH == 2 ? 4 : 8;
    const uint32_t seg = blockIdx.x * SEGS + q;
    uint32_t p, N;
    bool active = seg_packet(A.S, seg, p, N);
    constexpr uint32_t chanBits = DEPTH - 8 * bytes_shifted(DEPTH) + (CH == 2 ? 1 : 0);
    constexpr bool WIDE = chanBits > 17;
    const uint32_t frameBytes = A.S.frameSize * CH * bytes_per_sample(DEPTH);

    int na = 4, best = 0;
    if (active) {
        const PacketRec *rec = A.recs + p;
        if (rec->escape) active = false;  // escape estimate: the final pass does not run (:463)
        na = rec->c[c].num;
        best = (int)rec->mixRes;
    }
    LmsLane L = make_lane(lane, na, active ? (int)N : 0);
    int16_t *row = A.state + (uint64_t)seg * 64 + c * 32 + (na == 8 ? 16 : 0);
    int32_t a = (active && L.k < na) ? (int32_t)row[L.k] : 0;
    const uint32_t maxN = wave_max(active ? N : 0);
    const uint32_t chain = seg * CH + c;

    for (int j0 = 0; j0 < (int)maxN; j0 += kTile) {
        constexpr int ROWS = CH == 2 ? 4 : 8;
#pragma unroll
        for (int xr = 0; xr < ROWS; xr++) {
            constexpr int lanesPerSeg = CH == 2 ? 16 : 8;
            const uint32_t pq = (uint32_t)__builtin_amdgcn_readlane((int)p, xr * lanesPerSeg);
            const uint32_t Nq = (uint32_t)__builtin_amdgcn_readlane((int)N, xr * lanesPerSeg);
            const int bq = __builtin_amdgcn_readlane(best, xr * lanesPerSeg);
            const uint8_t *pk = A.S.pcm + (uint64_t)pq * frameBytes;
            if constexpr (CH == 2)
                stage_rows<DEPTH, 2>(xs, 2 * xr, 2 * xr + 1, pk, Nq, bq, j0, lane);
            else
                stage_rows<DEPTH, 1>(xs, xr, xr, pk, Nq, 0, j0, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        stage_p(xs + g * kXsStride, ps + g * kResStride, L.na, L.k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const int jEnd = min(j0 + kTile, (int)((maxN + 7) & ~7u));
        run_tile<WIDE>(a, xs + g * kXsStride, ps + g * kResStride, res + g * kResStride, L, j0, jEnd, chanBits);
        if (j0 == 0) warmup_fix(xs + g * kXsStride, res + g * kResStride, L, 32 - chanBits);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (active) flush_tile(res + g * kResStride, A.resC, A.chainsPad, chain, j0, N, L.k);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    if (active && L.k < na) row[L.k] = (int16_t)a;
}

// ================================================================================================
// Golomb kernels: one lane per stream, residuals read coalesced from the [sample][stream] planes
// ================================================================================================

// search1 counts: stream = r * chainsPad + chain, n8 symbols each -> bits1[stream]
template <int CH>
__global__ __launch_bounds__(64) void k_gol_count1(V1Args A, uint32_t chanBits)
{
    __shared__ uint32_t recip[17];
    gol_table_init(recip, threadIdx.x);
    __syncthreads();
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    const uint32_t chain = t % A.chainsPad, r = t / A.chainsPad;
    uint32_t p, N;
    const bool active = (r <= (uint32_t)kMaxRes) && seg_packet(A.S, chain / CH, p, N);
    const uint32_t n8 = active ? N / 8 : 0;
    const int32_t *src = A.resA + (uint64_t)r * A.chainsPad + chain;
    const uint64_t stride = 5ull * A.chainsPad;
    GolF g;
    golf_reset(g);
    golf_stream<false>(g, n8, wave_max(n8), chanBits, recip, [&](uint32_t j) { return src[j * stride]; });
    if (active) A.bits1[t] = g.bits;
}

// codec/ALACEncoder.cu:374-380: first minimum of bits1 + bits2 over mixRes 0..4
__global__ void k_decide1(V1Args A)
{
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t p, N;
    if (!seg_packet(A.S, seg, p, N)) return;
    uint32_t best = 0, minb = 1u << 31;
    for (uint32_t r = 0; r <= (uint32_t)kMaxRes; r++) {
        const uint32_t tot = A.bits1[r * A.chainsPad + seg * 2] + A.bits1[r * A.chainsPad + seg * 2 + 1];
        if (tot < minb) {
            minb = tot;
            best = r;
        }
    }
    A.recs[p].mixRes = best;
}

// search2 counts: stream = rowsel * chainsPad + chain.  Stereo: positions < P2 come from the last converge
// pass (resB), the tail from the mixRes = 4 search pass (resA) — codec/ALACEncoder.cu:433-445.
template <int CH>
__global__ __launch_bounds__(64) void k_gol_count2(V1Args A, uint32_t chanBits)
{
    __shared__ uint32_t recip[17];
    gol_table_init(recip, threadIdx.x);
    __syncthreads();
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    const uint32_t chain = t % A.chainsPad, rs = t / A.chainsPad;
    uint32_t p, N;
    const bool active = (rs <= 1) && seg_packet(A.S, chain / CH, p, N);
    const uint32_t n8 = active ? N / 8 : 0, n32 = N / 32, na = rs ? 8 : 4;
    uint32_t P2 = n8;
    if (CH == 2) {
        P2 = n32 > na + 1 ? n32 : na + 1;
        P2 = P2 < n8 ? P2 : n8;
    }
    const int32_t *srcB = A.resB + (uint64_t)rs * A.chainsPad + chain;
    const uint64_t strideB = 2ull * A.chainsPad;
    const int32_t *srcA = A.resA + (uint64_t)kMaxRes * A.chainsPad + chain;
    const uint64_t strideA = 5ull * A.chainsPad;
    GolF g;
    golf_reset(g);
    golf_stream<false>(g, n8, wave_max(n8), chanBits, recip,
                       [&](uint32_t j) { return j < P2 ? srcB[j * strideB] : srcA[j * strideA]; });
    if (active) A.cost2[t] = g.bits * 8 + 16 * na;  // :438, :447 / :899
}

// numU / numV, escape estimate (codec/ALACEncoder.cu:438-461, mono :899-915), header coefficients
template <int DEPTH, int CH>
__global__ void k_decide2(V1Args A)
{
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t p, N;
    if (!seg_packet(A.S, seg, p, N)) return;
    PacketRec *rec = A.recs + p;
    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    const uint32_t partial = (N != A.S.frameSize);
    uint32_t minBits = 0;
    for (uint32_t c = 0; c < (uint32_t)CH; c++) {
        const uint32_t chain = seg * CH + c;
        const uint32_t c4 = A.cost2[chain], c8 = A.cost2[A.chainsPad + chain];
        const uint32_t num = c8 < c4 ? 8 : 4;
        minBits += c8 < c4 ? c8 : c4;
        rec->c[c].num = (uint16_t)num;
        rec->c[c].bits = 0;
        const int16_t *row = A.state + (uint64_t)seg * 64 + c * 32 + (num == 8 ? 16 : 0);
        for (uint32_t k = 0; k < 8; k++) rec->c[c].coefs[k] = k < num ? row[k] : (int16_t)0;  // :477-485
    }
    if (CH == 1) {
        rec->c[1].num = 0;
        rec->c[1].bits = 0;
        rec->mixRes = 0;
    }
    minBits += (CH == 2 ? 64 : 32) + (partial ? 32 : 0) + N * (SHB * 8) * CH;
    const uint32_t escapeBits = N * DEPTH * CH + (partial ? 32 : 0) + 16;
    rec->numSamples = N;
    rec->escape = minBits >= escapeBits ? 1u : 0u;
}

// final entropy coding, one lane per chain (codec/ALACEncoder.cu:515-531, :944-945)
template <int CH>
__global__ __launch_bounds__(64) void k_gol_final(V1Args A, uint32_t chanBits)
{
    __shared__ uint32_t recip[17];
    gol_table_init(recip, threadIdx.x);
    __syncthreads();
    const uint32_t chain = blockIdx.x * 64u + threadIdx.x;
    uint32_t p, N;
    bool active = seg_packet(A.S, chain / CH, p, N);
    PacketRec *rec = A.recs + p;
    if (active && rec->escape) active = false;
    const uint32_t c = chain % CH;
    const uint32_t n = active ? N : 0;
    const int32_t *src = A.resC + chain;
    const uint64_t stride = A.chainsPad;
    GolF g;
    golf_reset(g);
    g.wp = A.bitWords + ((uint64_t)p * 2 + c) * A.wcap;
    g.wcap = A.wcap;
    golf_stream<true>(g, n, wave_max(n), chanBits, recip, [&](uint32_t j) { return src[j * stride]; });
    golf_flush<true>(g);
    if (active) rec->c[c].bits = g.bits;
}

// packet size + the post-hoc "compressed >= escape -> escape" rule (codec/ALACEncoder.cu:537-543, :952-958)
template <int DEPTH, int CH>
__global__ void k_finalize(PacketRec *recs, uint32_t *packetBytes, uint32_t numPackets, uint32_t frameSize)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= numPackets) return;
    PacketRec *rec = recs + p;
    constexpr uint32_t SHB = bytes_shifted(DEPTH);
    const uint32_t N = rec->numSamples;
    const uint32_t partial = (N != frameSize);
    const uint32_t escapeBits = N * DEPTH * CH + (partial ? 32 : 0) + 16;
    uint32_t body = 0;
    bool esc = rec->escape != 0;
    if (!esc) {
        body = 12 + 4 + (partial ? 32 : 0) + 16 + N * (SHB * 8) * CH;
        for (uint32_t c = 0; c < (uint32_t)CH; c++) body += 16 + 16 * rec->c[c].num + rec->c[c].bits;
        if (body >= escapeBits) esc = true;
    }
    if (esc) body = 12 + 4 + (partial ? 32 : 0) + N * DEPTH * CH;
    rec->escape = esc ? 1u : 0u;
    rec->totalBits = 7 + body + 3;
    packetBytes[p] = (7 + body + 3 + 7) / 8;
}

// init_coefs for every row of the working state (codec/ALACEncoder.cu:1524-1531)
__global__ void k_init_state(int16_t *state, uint32_t numSegments)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numSegments * 64u) return;
    const uint32_t k = i & 15;
    state[i] = (int16_t)(k == 0 ? 1216 : k == 1 ? -928 : k == 2 ? -64 : 0);
}

// ================================================================================================
// launcher
// ================================================================================================
template <int DEPTH, int CH>
static void launch_v1_typed(const V1Args &A0, uint32_t numPackets, uint32_t maxSegPackets, hipStream_t st,
                            hipEvent_t *ev, const PackArgs &pa)
{
    V1Args A = A0;
    const uint32_t nseg = A.S.numSegments;
    constexpr uint32_t chanBits = DEPTH - 8 * bytes_shifted(DEPTH) + (CH == 2 ? 1 : 0);
    const uint32_t streams = A.chainsPad;
    // stage events are recorded around the last packet position (the only one when every packet is
    // its own segment)
    for (uint32_t pos = 0; pos < maxSegPackets; pos++) {
        A.S.pos = pos;
        hipEvent_t *e = (ev && pos + 1 == maxSegPackets) ? ev : nullptr;
        if (e) (void)hipEventRecord(e[kStageLms1], st);
        if (CH == 2) hipLaunchKernelGGL(k_lms_search1<DEPTH>, dim3((nseg + 3) / 4), dim3(64), 0, st, A);
        if (e) (void)hipEventRecord(e[kStageGol1], st);
        if (CH == 2) {
            hipLaunchKernelGGL(k_gol_count1<CH>, dim3(5 * streams / 64), dim3(64), 0, st, A, chanBits);
            hipLaunchKernelGGL(k_decide1, dim3((nseg + 255) / 256), dim3(256), 0, st, A);
        }
        if (e) (void)hipEventRecord(e[kStageLms2], st);
        constexpr uint32_t S2 = CH == 2 ? 2 : 4;
        hipLaunchKernelGGL((k_lms_search2<DEPTH, CH>), dim3((nseg + S2 - 1) / S2), dim3(64), 0, st, A);
        if (e) (void)hipEventRecord(e[kStageGol2], st);
        hipLaunchKernelGGL(k_gol_count2<CH>, dim3(2 * streams / 64), dim3(64), 0, st, A, chanBits);
        hipLaunchKernelGGL((k_decide2<DEPTH, CH>), dim3((nseg + 255) / 256), dim3(256), 0, st, A);
        if (e) (void)hipEventRecord(e[kStageLms3], st);
        constexpr uint32_t S3 = CH == 2 ? 4 : 8;
        hipLaunchKernelGGL((k_lms_final<DEPTH, CH>), dim3((nseg + S3 - 1) / S3), dim3(64), 0, st, A);
        if (e) (void)hipEventRecord(e[kStageGol3], st);
        hipLaunchKernelGGL(k_gol_final<CH>, dim3(streams / 64), dim3(64), 0, st, A, chanBits);
    }
    if (ev) (void)hipEventRecord(ev[kStageScan], st);
    hipLaunchKernelGGL((k_finalize<DEPTH, CH>), dim3((numPackets + 255) / 256), dim3(256), 0, st, A.recs, A.packetBytes,
                       numPackets, A.S.frameSize);
    launch_scan_pack(DEPTH, CH, A.packetBytes, pa, numPackets, st, ev, false);
}

hipError_t launch_encode_v1(uint32_t depth, uint32_t channels, const EncodeArgs &ea, const PackArgs &pa,
                            const V1Buffers &vb, uint32_t numPackets, uint32_t maxSegPackets, hipStream_t st,
                            hipEvent_t *ev)
{
    V1Args A;
    A.S.pcm = ea.pcm;
    A.S.numSamples = ea.numSamples;
    A.S.segFirst = ea.segFirst;
    A.S.numSegments = ea.numSegments;
    A.S.frameSize = ea.frameSize;
    A.S.pos = 0;
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
    A.packetBytes = ea.packetBytes;
    if (!vb.stateInitialised)
        hipLaunchKernelGGL(k_init_state, dim3((ea.numSegments * 64 + 255) / 256), dim3(256), 0, st, vb.state,
                           ea.numSegments);
#define V1_CASE(D)                                                                                   \
    case D:                                                                                          \
        if (channels == 2)                                                                           \
            launch_v1_typed<D, 2>(A, numPackets, maxSegPackets, st, ev, pa);                         \
        else                                                                                         \
            launch_v1_typed<D, 1>(A, numPackets, maxSegPackets, st, ev, pa);                         \
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
