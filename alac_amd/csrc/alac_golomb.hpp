// alac_golomb.hpp — adaptive Golomb coder (dyn_comp, codec/ag_enc.c:249-367), streaming lane-serial
// form tuned for gfx950: one lane codes one stream; all 64 lanes of a wave walk their streams in
// lockstep over the sample index.
//
// What differs from the plain restatement in alac_dev.hpp (same bits out):
//   * the symbol's quotient n / m with m = 2^k - 1 is table-free: two add-and-shift rounds (golf_sym), exact on the only
//     branch that uses it (n < 9 m); only the rare run-length code (golf_close_run) still takes a multiply-high by the
//     tabulated reciprocal ceil(2^32 / m) — exact there because nz < 2^16 and m < 2^8;
//   * pb is the encoder's constant 40 (= (pbFactor 4 * PB0 40) / 4, codec/ALACEncoder.cu:365,515), so
//     pb * x is two shifts and an add instead of a quarter-rate 32-bit multiply;
//   * "numBits > 25 -> escape" (ag_enc.c:167) cannot fire for kb <= 14 (div <= 8, k <= 14) and is dropped.
#pragma once

#include "alac_dev.hpp"

namespace alacdev {

// recip[k] = ceil(2^32 / (2^k - 1)) for k = 2..16; k = 1 (m = 1) is handled by a select
__device__ __forceinline__ void gol_table_init(uint32_t *recip, int tid)
{
    if (tid < 17) {
        const uint64_t m = (1ull << tid) - 1;
        recip[tid] = tid >= 2 ? (uint32_t)(((1ull << 32) + m - 1) / m) : 0u;
    }
}

struct GolF {
    uint32_t mb, zmode, inrun, nz1, bits;  // nz1 = zeros swallowed by the open run + 1 (so that ">> 16" is the cap test)
    uint64_t acc;      // the most recent bits, right aligned; the low `nacc` of them are not yet a full word
    uint32_t nacc, wleft;
    uint32_t *wp;      // where the word being assembled goes
    uint32_t *wlim;    // !LAZY: wp is pulled back to this once per 16-symbol block (golf_stream) — the capacity guard
    // LAZY form only: completed words not yet stored — the last qn of (q0, q1, q2, q3), oldest first — so that they leave
    // four at a time in one 16-byte store (wp stays 16-byte aligned: slots are wcap words apart, wcap a multiple of 4)
    uint32_t q0, q1, q2, q3, qn;
};

__device__ __forceinline__ void golf_reset(GolF &g)
{
    g.mb = kMB0;
    g.zmode = 0;
    g.inrun = 0;
    g.nz1 = 1;
    g.bits = 0;
    g.acc = 0;
    g.nacc = 0;
    g.wleft = 0;
    g.wp = nullptr;
    g.wlim = nullptr;
    g.q0 = g.q1 = g.q2 = g.q3 = g.qn = 0;
}

// Append the low `nbits` (<= 32) of value.  The word under assembly is stored EVERY time (left aligned; a later
// put completes it in place) and the pointer advances when it fills up:
// no branch, no exec juggling — a lone wave pays ~16-24 cycles per branch, more than the redundant store.
// Only lanes that own a valid `wp` may call this (the callers sit inside the active-lane regions).
template <bool WRITE, bool LAZY = false>
__device__ __forceinline__ void golf_put(GolF &g, uint32_t value, uint32_t nbits)
{
    // WRITE: the bit count is where the word pointer ended up (golf_written_bits), not an add per symbol
    if constexpr (!WRITE) g.bits += nbits;
    if constexpr (WRITE) {
        g.acc = (g.acc << nbits) | (uint64_t)value;  // callers hand in values already confined to nbits
        g.nacc += nbits;
        // Latency regime (!LAZY): unconditional store — no branch, no exec juggling; the redundant stores are free when
        // a wave has its SIMD to itself (a predicated store measured +18 % on the whole final launch at 10 000 packets).
        // Throughput regime (LAZY): each of these stores is 64 lanes in 64 different cache lines, the slowest access
        // shape there is, and with many waves per CU they queue up behind each other: store only the ~0.4 words per
        // symbol that are complete (final coder at 125 000 packets: 4.95 -> 2.18 ms).
        const uint32_t word = (uint32_t)((g.acc << ((64u - g.nacc) & 63u)) >> 32);
        if constexpr (LAZY) {
            // a put completes at most one word (nacc < 32 before, nbits <= 32): shift it into the 4-word queue, and when
            // the queue is full send it off as ONE 16-byte store — a quarter of the scattered store instructions
            const bool full = (g.nacc >> 5) != 0;
            g.q0 = full ? g.q1 : g.q0;
            g.q1 = full ? g.q2 : g.q1;
            g.q2 = full ? g.q3 : g.q2;
            g.q3 = full ? word : g.q3;
            g.qn += full ? 1u : 0u;
            if (g.qn == 4) {
                *(uint4 *)g.wp = make_uint4(g.q0, g.q1, g.q2, g.q3);
                const uint32_t adv = g.wleft >= 4 ? 4u : 0u;  // capacity reached: stay (the packet escapes anyway)
                g.wleft -= adv;
                g.wp += adv;
                g.qn = 0;
            }
        } else {
            // no capacity test per symbol: a put completes at most one word, a symbol makes at most two puts, and
            // golf_stream pulls wp back to wlim (34 words before the slot's end) once per 16 symbols
            *g.wp = word;
            g.wp += g.nacc >> 5;
        }
        g.nacc &= 31u;
    }
}

// Start a written stream in its slot of `wcap` words.  A stream that reaches the guard (wcap - 34 words, resp. the last
// 16-byte group in the LAZY form) keeps overwriting the slot's tail; enc_layout sizes wcap so that such a stream is
// longer than the escape size by itself: the packet is sent uncompressed and nobody reads the slot.
__device__ __forceinline__ void golf_open(GolF &g, uint32_t *slot, uint32_t wcap)
{
    g.wp = slot;
    g.wlim = slot + (wcap - 34);
    g.wleft = wcap - 1;
}

// bits of a written stream (exact unless the guard was reached, and then at least 32 (wcap - 34))
template <bool LAZY = false>
__device__ __forceinline__ uint32_t golf_written_bits(const GolF &g, const uint32_t *slot)
{
    return 32u * ((uint32_t)(g.wp - slot) + (LAZY ? g.qn : 0u)) + g.nacc;
}

// the bits left over after the last full word (golf_put stores before it advances)
template <bool WRITE, bool LAZY = false>
__device__ __forceinline__ void golf_flush(GolF &g)
{
    if constexpr (WRITE) {
        if constexpr (LAZY) {
            // queued whole words first (the last qn of q0..q3), then the partial one behind them
            if (g.qn == 3) g.wp[0] = g.q1;
            if (g.qn >= 2) g.wp[g.qn - 2] = g.q2;
            if (g.qn >= 1) g.wp[g.qn - 1] = g.q3;
            if (g.nacc > 0) g.wp[g.qn] = (uint32_t)((g.acc << (64u - g.nacc)) >> 32);
        } else {
            if (g.nacc > 0) *g.wp = (uint32_t)((g.acc << (64u - g.nacc)) >> 32);
        }
    }
}

// run-length code (dyn_code, ag_enc.c:115-148) then mb = 0 (:351-358); k is 2..8 here
template <bool WRITE, bool LAZY = false>
__device__ __forceinline__ void golf_close_run(GolF &g, const uint32_t *recip)
{
    const uint32_t k = (uint32_t)(lead(g.mb) - 24 + (int32_t)((g.mb + 16u) >> 6));
    const uint32_t mz = (1u << k) - 1;  // & wb is the identity: k <= 8 < kb
    const uint32_t nz = g.nz1 - 1;
    const uint32_t div = __umulhi(nz, recip[k]);  // nz < 2^16, mz < 2^8
    uint32_t numBits, value;
    if (div >= kMaxPrefix) {
        numBits = kMaxPrefix + kMaxRunBits;
        value = (((1u << kMaxPrefix) - 1) << kMaxRunBits) + nz;
    } else {
        const uint32_t mod = nz - div * mz;
        const uint32_t de = (mod == 0);
        numBits = div + k + 1 - de;
        value = (((1u << div) - 1) << (numBits - div)) + mod + 1 - de;
    }
    golf_put<WRITE, LAZY>(g, value, numBits);
    g.mb = 0;
    g.inrun = 0;
}

// one residual.  CHECKED: `valid` = this lane still has samples (only needed in blocks where some lane of the
// wave has run out).  The end of the stream is NOT handled here: a run still open after the last residual is
// closed by golf_finish, and a run "entered" by the last residual is dropped there (ag_enc.c:328 enters zero
// mode only when c < numSamples).
// Control flow is kept to three short regions so that a wave whose lanes are in different coder states
// (zero run / normal / escape) does not execute long divergent bodies: (A) run bookkeeping, (B) run close,
// entered only when some lane closes a run, (C) the symbol itself, straight-line (escape handled by selects).
// Requires bitSize <= 23 so that the escape (9 ones + bitSize raw bits) is one <= 32-bit put.
// ZZ: the plane already holds the zig-zag image of the residuals (the final predictor pass stores it that way:
// the map is three instructions per residual here, three per 64 residuals in the predictor's row flush)
template <bool WRITE, bool CHECKED, bool ZZ = false, bool LAZY = false>
__device__ __forceinline__ void golf_sym(GolF &g, int32_t del, bool valid, uint32_t bitSize, const uint32_t *recip)
{
    // (A) ag_enc.c:333-349, on 0/1 flags in vector registers (boolean chains through scalar masks cost a lone
    // wave three times as many instructions): del == 0 <=> its zig-zag image is 0
    const uint32_t t2 = ZZ ? (uint32_t)del : (((uint32_t)del << 1) ^ (uint32_t)(del >> 31));  // n + zmode = 2|del| - (del < 0)
    uint32_t nzf;                                                      // del != 0
    asm("v_min_u32 %0, 1, %1" : "=v"(nzf) : "v"(t2));                  // (one instruction; the compiler prefers compare + select)
    const uint32_t inr = CHECKED ? (valid ? g.inrun : 0u) : g.inrun;
    const uint32_t sw = inr & (nzf ^ 1u);                              // a zero swallowed by the open run
    g.nz1 += sw;
    // the run ends before this residual (non-zero) or at it (it just reached 65535 zeros: nz1 = 65536; an open run
    // never holds more, so the shift is 0 whenever the residual is non-zero)
    const uint32_t cl = inr & (nzf | (g.nz1 >> 16));
    // (B) one predicated region for the (rare) close
    if (cl) {
        const uint32_t capf = sw;  // closed at a zero: the cap
        golf_close_run<WRITE, LAZY>(g, recip);
        if (capf) g.zmode = 0;
    }
    // keep (C) ONE copy behind the join: left alone, the compiler threads "mb = 0 after a close" into a second copy
    // of the k computation and pays for it with an exec-mask dance on the path that skips the close
    asm volatile("" : "+v"(g.mb));
    // (C) ag_enc.c:285-331.  Where every lane of the wave is inside its stream (!CHECKED) the symbol is computed
    // by ALL lanes and a lane that swallowed a zero simply keeps its state and appends nothing: an exec-masked
    // region costs a lone wave a save/branch/restore sequence worth a dozen instructions, the selects cost six.
    const bool live = CHECKED ? (valid && !sw) : !sw;
    if (CHECKED ? live : true) {
        // k = min(lg3a(mb >> 9), kb); lg3a(x) = 31 - clz(x + 3) and clz((mb >> 9) + 3) = clz(mb + 1536) + 9
        const uint32_t k = min(22u - (uint32_t)__builtin_clz(g.mb + (3u << kQBShift)), kKB0);
        const uint32_t m = (1u << k) - 1;
        const uint32_t n = t2 - g.zmode;
        const bool esc = n >= m * 9;  // div >= MAX_PREFIX_32
        // n / m for m = 2^k - 1, needed only when n < 9 m (d = n / m <= 8): two rounds of "add the previous estimate and
        // shift" — d1 = (n + (n >> k) + 1) >> k, div = (n + d1 + 1) >> k — five instructions, no table, no division, no
        // k == 1 case.  Since n + 1 = d 2^k + (r + 1 - d) with 0 <= r < m, adding an estimate e of d in [d - 1, d] lands
        // in [d 2^k, (d + 1) 2^k); n >> k is within 1 of d for k >= 3 and the second round repairs k = 2.  For k = 1
        // (m = 1) div can still come out one short at n = 7, 8 — and the code is the SAME: with m = 1 a short div only
        // moves a one from the unary part into the remainder field.  Every (k, n) of the range is enumerated by
        // tests/test_coder_division.py (quotient for k >= 2, emitted (numBits, value) for k >= 1).
        // (Round 3: this replaced a two-level shift-and-add with a select for k = 1: thirteen instructions.)
        const uint32_t d1 = (n + (n >> k) + 1u) >> k;
        const uint32_t div = (n + d1 + 1u) >> k;
        const uint32_t mod = n - __umul24(div, m);
        // dyn_code_32bit (:151-183): numBits = div + k + 1 - [mod == 0], value = ones(div) 0 (mod + 1 - [mod == 0])
        const uint32_t ne = min(mod, 1u);
        const uint32_t kd = k + ne;
        uint32_t numBits = div + kd;
        uint32_t value = (((1u << div) - 1) << kd) + mod + ne;
        if (esc) {
            numBits = kMaxPrefix + bitSize;
            value = (((1u << kMaxPrefix) - 1) << bitSize) | (n & ((1u << bitSize) - 1));
        }
        if (!live) {
            numBits = 0;
            value = 0;
        }
        golf_put<WRITE, LAZY>(g, value, numBits);
        // mb = pb * (n + zmode) + mb - ((pb * mb) >> 9), pb = 40   (:318)
        // (mb can pass 2^24 under sustained large residuals: shifts, not a 24-bit multiply; t2 < 2^24 always)
        // (40 mb) >> 9 == (5 mb) >> 6, and 5 mb < 2^32 for every reachable mb (< 2^26)
        uint32_t mb = __umul24(t2, 40u) + g.mb - (times5(g.mb) >> (kQBShift - 3));
        mb = n > kMeanClamp ? kMeanClamp : mb;
        const bool enter = mb < (1u << (kQBShift - 2));  // (mb << 2) < QB, :328
        g.mb = live ? mb : g.mb;
        g.zmode = live ? (enter ? 1u : 0u) : g.zmode;
        g.inrun = live ? (enter ? 1u : 0u) : g.inrun;
        g.nz1 = (live && enter) ? 1u : g.nz1;
    }
}

// after the last residual: a run that swallowed zeros up to the end is coded now (:351); one that was only
// just entered by the last residual does not exist in the reference (:328, c < numSamples) and is dropped
template <bool WRITE, bool LAZY = false>
__device__ __forceinline__ void golf_finish(GolF &g, bool active, const uint32_t *recip)
{
    const bool close = active && g.inrun && g.nz1 > 1;
    if (__any(close)) {
        if (close) golf_close_run<WRITE, LAZY>(g, recip);
    }
    g.inrun = 0;
}

// Walk one stream of `n` residuals.  `fetch(j)` returns this lane's residual of row j, with j WAVE-UNIFORM: every
// lane loads every row up to the wave's longest stream (rows beyond a lane's own length are ignored, never
// branched around), so the address is a scalar row base plus the lane's column.  Three 16-sample register
// buffers rotate so that a block's loads are issued two blocks (32 symbols) before it is coded: the bit-word
// stores of the coder share the vector-memory counter with the loads, so the compiler can only wait for
// "everything outstanding" (vmcnt(0)) — with two blocks of distance that wait finds the loads done.
// `need(rows)` is called (wave-uniformly) before rows < `rows` of the plane are read: a no-op when the plane was
// written by an earlier kernel, a flag wait + acquire when a producer in the same launch is still writing it.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

struct NoWait {
    __device__ __forceinline__ void operator()(uint32_t) const {}
};

// Row source of golf_stream: rows below `split` come from plane p0 (row stride s0 elements), the rest from p1/s1;
// the lane reads column `col` of a row.  Row addresses are scalar and advance by one add per row (no multiply).
struct RowSrc {
    const int32_t *p0, *p1;
    uint64_t s0, s1;
    uint32_t split;  // 0xffffffff: single plane p0
    uint32_t col;
};

__device__ __forceinline__ RowSrc one_plane(const int32_t *plane, uint64_t stride, uint32_t col)
{
    RowSrc r;
    r.p0 = r.p1 = plane;
    r.s0 = r.s1 = stride;
    r.split = 0xffffffffu;
    r.col = col;
    return r;
}

// finish (per lane): the stream ends with these `n` residuals — a run still open is coded (golf_finish).  false where the
// residuals continue elsewhere (the first part of a coder split over two waves, k_final_fused).
template <bool WRITE, bool ZZ = false, class Need = NoWait, bool LAZY = false>
__device__ __forceinline__ void golf_stream(GolF &g, uint32_t n, uint32_t nMaxWave, uint32_t bitSize,
                                            const uint32_t *recip, const RowSrc &R, Need &&need = Need(),
                                            bool idleFast = true, bool finish = true)
{
    constexpr int B = 16;
    int32_t bufA[B], bufB[B], bufC[B];
    auto load = [&](int32_t (&buf)[B], uint32_t jb) {
        if (jb < nMaxWave) {
            need(min(jb + B, nMaxWave));
            if (jb + B <= R.split || jb >= R.split) {
                const bool first = jb + B <= R.split;
                const uint64_t st = first ? R.s0 : R.s1;
                // scalar row base + the lane's 32-bit byte offset: a global load with an SGPR base and a VGPR offset, no
                // 64-bit vector address arithmetic per row
                const char *base = (const char *)((first ? R.p0 : R.p1) + (uint64_t)jb * st);
                const uint32_t rowBytes = (uint32_t)st * 4u;  // < 2^32: a row is at most 5 planes x chains x 4 bytes
                uint32_t off = R.col * 4u;                     // 16 rows stay below 2^32 bytes as well
#pragma unroll
                for (int s = 0; s < B; s++) {
                    buf[s] = *(const int32_t *)(base + off);
                    off += rowBytes;
                }
            } else {  // the block straddles the split (never with 4096-sample packets: split = 128)
#pragma unroll
                for (int s = 0; s < B; s++) {
                    const uint32_t j = jb + s;
                    buf[s] = (j < R.split ? R.p0 + (uint64_t)j * R.s0 : R.p1 + (uint64_t)j * R.s1)[R.col];
                }
            }
        }
    };
    // lanes without a stream (n = 0: pad lanes, escape packets) do not count: in the unchecked blocks they code
    // whatever they load into state and bit words nobody reads (their wp points at a spare slot)
    const uint32_t nMinWave = wave_min_u32(n ? n : (idleFast ? 0xffffffffu : 0u));
    // Two copies of the block loop instead of a per-block choice inside one: the unchecked body (what runs for all
    // but the last blocks of a wave) then fits the instruction cache — 48 unrolled symbols are ~36 KB, and with
    // the checked variant in the same loop the body would be twice that (a 3 x 32-symbol body, 147 KB, was measured
    // 3.5 x slower: the 64 KB instruction cache is a hard limit for these kernels).
    auto guard = [&]() {
        if constexpr (WRITE && !LAZY) g.wp = g.wp < g.wlim ? g.wp : g.wlim;
    };
    auto codeFast = [&](const int32_t (&buf)[B]) {
        guard();
#pragma unroll
        for (int s = 0; s < B; s++) golf_sym<WRITE, false, ZZ, LAZY>(g, buf[s], true, bitSize, recip);
    };
    auto codeChecked = [&](const int32_t (&buf)[B], uint32_t jb) {
        if (jb >= nMaxWave) return;
        guard();
#pragma unroll
        for (int s = 0; s < B; s++) golf_sym<WRITE, true, ZZ, LAZY>(g, buf[s], jb + s < n, bitSize, recip);
    };
    const uint32_t fastEnd = (min(nMinWave, nMaxWave) / (3 * B)) * (3 * B);  // whole iterations every lane fully owns
    load(bufA, 0);
    load(bufB, B);
    uint32_t jb = 0;
    for (; jb < fastEnd; jb += 3 * B) {
        load(bufC, jb + 2 * B);
        codeFast(bufA);
        load(bufA, jb + 3 * B);
        codeFast(bufB);
        load(bufB, jb + 4 * B);
        codeFast(bufC);
    }
    for (; jb < nMaxWave; jb += 3 * B) {
        load(bufC, jb + 2 * B);
        codeChecked(bufA, jb);
        load(bufA, jb + 3 * B);
        codeChecked(bufB, jb + B);
        load(bufB, jb + 4 * B);
        codeChecked(bufC, jb + 2 * B);
    }
    const uint32_t openRun = g.inrun;
    golf_finish<WRITE, LAZY>(g, n > 0 && finish, recip);
    if (!finish) g.inrun = openRun;  // the run continues in the caller's next part
}

// functor form (any per-lane row choice), used only where lanes of one wave disagree about the planes
template <bool WRITE, class Fetch, class Need = NoWait>
__device__ __forceinline__ void golf_stream_fn(GolF &g, uint32_t n, uint32_t nMaxWave, uint32_t bitSize,
                                            const uint32_t *recip, Fetch &&fetch, Need &&need = Need(), bool idleFast = true,
                                            bool finish = true)
{
    constexpr int B = 16;
    int32_t bufA[B], bufB[B], bufC[B];
    auto load = [&](int32_t (&buf)[B], uint32_t jb) {
        if (jb < nMaxWave) {
            need(min(jb + B, nMaxWave));
#pragma unroll
            for (int s = 0; s < B; s++) buf[s] = fetch(jb + s);
        }
    };
    // lanes without a stream (n = 0: pad lanes, escape packets) do not count: in the unchecked blocks they code
    // whatever they load into state and bit words nobody reads (their wp points at a spare slot)
    const uint32_t nMinWave = wave_min_u32(n ? n : (idleFast ? 0xffffffffu : 0u));
    auto code = [&](const int32_t (&buf)[B], uint32_t jb) {
        if (jb >= nMaxWave) return;
        if (jb + B <= nMinWave) {  // every lane that has a stream owns the whole block
#pragma unroll
            for (int s = 0; s < B; s++) golf_sym<WRITE, false>(g, buf[s], true, bitSize, recip);
        } else {
#pragma unroll
            for (int s = 0; s < B; s++) golf_sym<WRITE, true>(g, buf[s], jb + s < n, bitSize, recip);
        }
    };
    load(bufA, 0);
    load(bufB, B);
    for (uint32_t jb = 0; jb < nMaxWave; jb += 3 * B) {
        load(bufC, jb + 2 * B);
        code(bufA, jb);
        load(bufA, jb + 3 * B);
        code(bufB, jb + B);
        load(bufB, jb + 4 * B);
        code(bufC, jb + 2 * B);
    }
    const uint32_t openRun = g.inrun;
    golf_finish<WRITE>(g, n > 0 && finish, recip);
    if (!finish) g.inrun = openRun;  // the caller finishes the stream
}

}  // namespace alacdev
