// alac_golomb.hpp — adaptive Golomb coder (dyn_comp, codec/ag_enc.c:249-367), streaming lane-serial
// form tuned for gfx950: one lane codes one stream; all 64 lanes of a wave walk their streams in
// lockstep over the sample index.
//
// What differs from the plain restatement in alac_dev.hpp (same bits out):
//   * n / m with m = 2^k - 1 is a multiply-high by a tabulated reciprocal ceil(2^32 / m): exact whenever
//     n * m < 2^32, which holds on the only branch that uses the quotient (n < 9 m, k <= 14);
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
    uint32_t mb, zmode, inrun, nz, bits;
    uint64_t acc;
    uint32_t nacc, widx, wcap;
    uint32_t *wp;
};

__device__ __forceinline__ void golf_reset(GolF &g)
{
    g.mb = kMB0;
    g.zmode = 0;
    g.inrun = 0;
    g.nz = 0;
    g.bits = 0;
    g.acc = 0;
    g.nacc = 0;
    g.widx = 0;
    g.wcap = 0;
    g.wp = nullptr;
}

template <bool WRITE>
__device__ __forceinline__ void golf_put(GolF &g, uint32_t value, uint32_t nbits)
{
    g.bits += nbits;
    if constexpr (WRITE) {
        g.acc = (g.acc << nbits) | (uint64_t)value;  // callers hand in values already confined to nbits
        g.nacc += nbits;
        if (g.nacc >= 32) {
            g.nacc -= 32;
            if (g.widx < g.wcap) g.wp[g.widx] = (uint32_t)(g.acc >> g.nacc);
            g.widx++;
        }
    }
}

template <bool WRITE>
__device__ __forceinline__ void golf_flush(GolF &g)
{
    if constexpr (WRITE) {
        if (g.nacc > 0) {
            if (g.widx < g.wcap) g.wp[g.widx] = (uint32_t)(g.acc << (32 - g.nacc));
            g.widx++;
            g.nacc = 0;
        }
    }
}

// run-length code (dyn_code, ag_enc.c:115-148) then mb = 0 (:351-358); k is 2..8 here
template <bool WRITE>
__device__ __forceinline__ void golf_close_run(GolF &g, const uint32_t *recip)
{
    const uint32_t k = (uint32_t)(lead(g.mb) - 24 + (int32_t)((g.mb + 16u) >> 6));
    const uint32_t mz = (1u << k) - 1;  // & wb is the identity: k <= 8 < kb
    const uint32_t nz = g.nz;
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
    golf_put<WRITE>(g, value, numBits);
    g.mb = 0;
    g.inrun = 0;
}

// one residual; `valid` = this lane still has samples, `last` = final sample of its block.
// Control flow is kept to three short regions so that a wave whose lanes are in different coder states
// (zero run / normal / escape) does not execute long divergent bodies: (A) run bookkeeping, (B) run close,
// entered only when some lane closes a run, (C) the symbol itself with the escape handled by selects.
// Requires bitSize <= 23 so that the escape (9 ones + bitSize raw bits) is one <= 32-bit put.
template <bool WRITE>
__device__ __forceinline__ void golf_sym(GolF &g, int32_t del, bool valid, bool last, uint32_t bitSize,
                                         const uint32_t *recip)
{
    // (A) ag_enc.c:333-349
    const bool inrun = valid && g.inrun;
    const bool swallow = inrun && (del == 0);
    g.nz += swallow ? 1u : 0u;
    const bool cap = swallow && (g.nz >= 65535);
    const bool close = inrun && (!swallow || cap || last);
    // (B)
    if (__any(close)) {
        if (close) {
            golf_close_run<WRITE>(g, recip);
            if (cap) g.zmode = 0;
        }
    }
    // (C) ag_enc.c:285-331
    if (valid && !swallow) {
        const uint32_t k = min((uint32_t)lg3a(g.mb >> kQBShift), kKB0);
        const uint32_t m = (1u << k) - 1;
        const uint32_t a = (uint32_t)(del < 0 ? -del : del);
        const uint32_t t2 = (a << 1) - ((uint32_t)del >> 31);  // n + zmode
        const uint32_t n = t2 - g.zmode;
        const bool esc = n >= m * 9;  // div >= MAX_PREFIX_32
        // n / m for m = 2^k - 1, needed only when n < 9 m (else escape).  n = q0 * 2^k + r0 = q0 * m + (q0 + r0)
        // with q0 <= 8, so for m >= 15 the quotient is q0 or q0 + 1; the three small moduli (1, 3, 7; n < 63)
        // use an 8-bit reciprocal (256, 86, 37: exact on that range).  No table, no division.
        const uint32_t q0 = n >> k;
        const uint32_t t0 = q0 + (n & m);
        const uint32_t divA = q0 + (t0 >= m ? 1u : 0u);
        const uint32_t c8 = k == 1 ? 256u : (k == 2 ? 86u : 37u);
        const uint32_t divB = __umul24(n & 0xffu, c8) >> 8;
        const uint32_t div = k >= 4 ? divA : divB;
        const uint32_t mod = n - __umul24(div, m);
        const uint32_t de = (mod == 0);
        uint32_t numBits = div + k + 1 - de;
        uint32_t value = (((1u << div) - 1) << (numBits - div)) + mod + 1 - de;
        if (esc) {
            numBits = kMaxPrefix + bitSize;
            value = (((1u << kMaxPrefix) - 1) << bitSize) | (n & ((1u << bitSize) - 1));
        }
        golf_put<WRITE>(g, value, numBits);
        // mb = pb * (n + zmode) + mb - ((pb * mb) >> 9), pb = 40   (:318)
        const uint32_t pm = (g.mb << 5) + (g.mb << 3);
        uint32_t mb = (t2 << 5) + (t2 << 3) + g.mb - (pm >> kQBShift);
        mb = n > kMeanClamp ? kMeanClamp : mb;
        const bool enter = (mb < (1u << (kQBShift - 2))) && !last;  // (mb << 2) < QB, :328
        g.mb = mb;
        g.zmode = enter ? 1u : 0u;
        g.inrun = enter ? 1u : 0u;
        g.nz = enter ? 0u : g.nz;
    }
}

// Walk one stream of `n` residuals laid out with `stride` (the [sample][stream] planes).  Three 16-sample
// register buffers rotate so that a block's loads are issued two blocks (32 symbols) before it is coded:
// the bit-word stores of the coder share the vector-memory counter with the loads, so the compiler can only
// wait for "everything outstanding" (vmcnt(0)) — with two blocks of distance that wait finds the loads done.
// `need(rows)` is called (wave-uniformly) before rows < `rows` of the plane are read: a no-op when the plane was
// written by an earlier kernel, a flag wait + acquire when a producer in the same launch is still writing it.
struct NoWait {
    __device__ __forceinline__ void operator()(uint32_t) const {}
};

template <bool WRITE, class Fetch, class Need = NoWait>
__device__ __forceinline__ void golf_stream(GolF &g, uint32_t n, uint32_t nMaxWave, uint32_t bitSize,
                                            const uint32_t *recip, Fetch &&fetch, Need &&need = Need())
{
    constexpr int B = 16;
    int32_t bufA[B], bufB[B], bufC[B];
    auto load = [&](int32_t (&buf)[B], uint32_t jb) {
        if (jb < nMaxWave) need(min(jb + B, nMaxWave));
#pragma unroll
        for (int s = 0; s < B; s++) buf[s] = (jb + s) < n ? fetch(jb + s) : 0;
    };
    auto code = [&](const int32_t (&buf)[B], uint32_t jb) {
        if (jb >= nMaxWave) return;
#pragma unroll
        for (int s = 0; s < B; s++) {
            const uint32_t j = jb + s;
            golf_sym<WRITE>(g, buf[s], j < n, j + 1 == n, bitSize, recip);
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
}

}  // namespace alacdev
