// alac_dev.hpp — device-side building blocks of the ALAC hot path for gfx950 (wave64).
//
// Integer-only code: int32 two's-complement with int16 coefficient storage, exactly the
// arithmetic of the reference stage files (cited per function, paths relative to the reference
// tree).  No floating point on any result-bearing path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alacdev {

// ---- codec constants: codec/aglib.h:36-52, codec/dplib.h:41, codec/ALACEncoder.cu:56-61 ----
constexpr uint32_t kQBShift = 9;
constexpr uint32_t kPB0 = 40;
constexpr uint32_t kMB0 = 10;
constexpr uint32_t kKB0 = 14;
constexpr uint32_t kDenShift = 9;
constexpr uint32_t kMaxPrefix = 9;      // MAX_PREFIX_16 == MAX_PREFIX_32
constexpr uint32_t kMaxRunBits = 16;    // MAX_DATATYPE_BITS_16
constexpr uint32_t kMeanClamp = 0xffffu;
constexpr int32_t kMixBits = 2;         // kDefaultMixBits
constexpr int32_t kMaxRes = 4;          // kMaxRes

__host__ __device__ constexpr uint32_t bytes_per_sample(uint32_t depth)
{
    return depth == 16 ? 2u : (depth == 32 ? 4u : 3u);
}
// codec/ALACEncoder.cu:327-332
__host__ __device__ constexpr uint32_t bytes_shifted(uint32_t depth)
{
    return depth == 32 ? 2u : (depth >= 24 ? 1u : 0u);
}

__device__ __forceinline__ int32_t sext(int32_t x, uint32_t chanshift)
{
    return (int32_t)((uint32_t)x << chanshift) >> chanshift;
}
__device__ __forceinline__ int32_t sign_of(int32_t x) { return min(max(x, -1), 1); }
// Two's-complement wrap-around made explicit.  The reference's int32 predictor arithmetic overflows for full-scale 32-bit
// material (chanBits 32: a difference of two samples has 33 significant bits) and its compiled objects wrap (the oracle pins
// exactly that, -fwrapv); signed overflow in device code is undefined, and the compiler uses that — sign_of(top - x) became
// a comparison of top with x.  Every generic predictor form (any chanBits) computes with these; the hot fast paths are
// restricted to chanBits <= 23, where no difference can overflow.
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__device__ __forceinline__ int32_t wabs(int32_t a) { return a < 0 ? wsub(0, a) : a; }
// x * 5 as one shift-add (the compiler turns (x << 2) + x back into a quarter-rate 32-bit multiply)
__device__ __forceinline__ uint32_t times5(uint32_t x)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(r) : "v"(x));
    return r;
}

// lead(): codec/ag_enc.c:65-77, lead(0) == 32
__device__ __forceinline__ int32_t lead(uint32_t m) { return m ? __clz((int)m) : 32; }
__device__ __forceinline__ int32_t lg3a(uint32_t x) { return 31 - lead(x + 3); }

// ---- PCM sample access ------------------------------------------------------------------------
// Full-precision sample t (channel-interleaved index) of a packet, sign-extended to int32:
// 16: int16; 20: (x<<8)>>12 of 3 LE bytes; 24: (x<<8)>>8; 32: int32
// (codec/matrix_enc.cu:79-82, :129-134, :197-202, :338-342).
template <int DEPTH>
__device__ __forceinline__ int32_t load_sample(const uint8_t *pk, uint32_t t)
{
    if constexpr (DEPTH == 16) {
        return ((const int16_t *)pk)[t];
    } else if constexpr (DEPTH == 32) {
        return ((const int32_t *)pk)[t];
    } else {
        const uint8_t *p = pk + 3u * t;
        uint32_t w = ((uint32_t)p[2] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 8);
        return (int32_t)w >> (DEPTH == 20 ? 12 : 8);
    }
}

// Left/right of stereo sample-frame j after the shift-off of the low bytes (what every mixNN
// kernel feeds the matrix; the shifted-off bytes are emitted by the packer straight from PCM).
template <int DEPTH>
__device__ __forceinline__ void load_lr(const uint8_t *pk, uint32_t j, int32_t &l, int32_t &r)
{
    constexpr int SH = 8 * (int)bytes_shifted(DEPTH);
    if constexpr (DEPTH == 16) {
        int32_t w = ((const int32_t *)pk)[j];
        l = (int16_t)w;
        r = w >> 16;
    } else {
        l = load_sample<DEPTH>(pk, 2 * j) >> SH;
        r = load_sample<DEPTH>(pk, 2 * j + 1) >> SH;
    }
}

// mix: codec/matrix_enc.cu:72-99 (and the 20/24/32-bit copies): channel 0 = u, channel 1 = v
__device__ __forceinline__ int32_t mix_sample(int32_t mixres, int ch, int32_t l, int32_t r)
{
    if (mixres != 0) {
        int32_t m2 = (1 << kMixBits) - mixres;
        return ch == 0 ? ((mixres * l + m2 * r) >> kMixBits) : (l - r);
    }
    return ch == 0 ? l : r;
}

// ---- adaptive FIR predictor, lane-serial form --------------------------------------------------
// One lane walks one chain.  pc_block's three code paths (codec/dp_enc.c:116-195 4-tap,
// :196-340 8-tap, :341-387 general) are the same recurrence; NA is a compile-time tap count so
// the history and coefficients stay in registers.
//   h[0] = in[j-1] ... h[NA] = in[j-NA-1] (= "top")
template <int NA>
struct Lms {
    int32_t a[NA];      // coefficients, kept sign-extended from int16
    int32_t h[NA + 1];  // input history
};

template <int NA>
__device__ __forceinline__ void lms_push(Lms<NA> &s, int32_t x)
{
#pragma unroll
    for (int k = NA; k > 0; k--) s.h[k] = s.h[k - 1];
    s.h[0] = x;
}

// The coefficient walk of dp_enc.c:143-188 / dp_dec.c:131-181, in the reference's own terms: del0 shrinks (grows, for a negative
// residual) by (NA - k) * ((+-sign(dd) * dd) >> denshift) tap by tap, and the walk STOPS for good the first time del0 has
// changed sign.  (Rounds 1-3 evaluated "tap k is touched iff |del| - the partial sum is still positive" per tap, which is the
// same thing while the partial sums only grow — and is not once they wrap: 32-bit mono material with a small denShift makes a
// single term exceed 2^31, the remainder turns positive again and a later tap got an update the reference had already
// broken out of.  Found by round 4's foreign-stream soak, seed 289.  The hot kernels have their own threshold forms and only
// run where chanBits <= 23 and denShift = 9: no term can wrap there.)
template <int NA>
__device__ __forceinline__ void lms_adapt(Lms<NA> &s, const int32_t (&b)[NA], int32_t del, uint32_t denshift)
{
    const int32_t sg = sign_of(del);
    int32_t del0 = del;
    bool go = sg != 0;
#pragma unroll
    for (int k = NA - 1; k >= 0; k--) {
        const int32_t sgn = sign_of(b[k]);           // b[k] = top - in[j - 1 - k] = dd
        const int32_t step = wmul(sg, sgn);          // sg > 0: coefs[k] -= sgn; sg < 0: coefs[k] += sgn
        s.a[k] = (int16_t)(s.a[k] - (go ? step : 0));
        del0 = wsub(del0, wmul(NA - k, wmul(step, b[k]) >> denshift));
        go = go && (sg > 0 ? del0 > 0 : del0 < 0);
    }
}

// One encoder step for j > NA: returns the residual and adapts the coefficients (sign-LMS).
// The early-exit update loop of dp_enc.c:143-188 is evaluated branch-free: with E = |del| and
// t_k = (|b_k| + (del < 0 ? 511 : 0)) >> 9  [ (sgn*b)>>9 resp. -((-sgn*b)>>9) ], tap k is
// touched iff E - sum_{i>k} (NA-i) t_i > 0, because the partial sums only grow.
template <int NA>
__device__ __forceinline__ int32_t lms_step_enc(Lms<NA> &s, int32_t x, uint32_t chanshift)
{
    const int32_t top = s.h[NA];
    int32_t b[NA];
    int32_t sum = 0;
#pragma unroll
    for (int k = 0; k < NA; k++) {
        b[k] = wsub(top, s.h[k]);
        sum = wadd(sum, wmul(s.a[k], b[k]));
    }
    int32_t del = sext(wsub(wsub(x, top), wsub(1 << (kDenShift - 1), sum) >> kDenShift), chanshift);
    lms_adapt<NA>(s, b, del, kDenShift);
    lms_push<NA>(s, x);
    return del;
}

// Decoder step (codec/dp_dec.c:116-182 / :204-324 / :338-379): same adaptation driven by the
// decoded residual; returns the reconstructed sample.
template <int NA>
__device__ __forceinline__ int32_t lms_step_dec(Lms<NA> &s, int32_t del, uint32_t chanshift,
                                                uint32_t denshift)
{
    const int32_t top = s.h[NA];
    int32_t b[NA];
    int32_t sum = 0;
#pragma unroll
    for (int k = 0; k < NA; k++) {
        b[k] = wsub(top, s.h[k]);
        sum = wadd(sum, wmul(s.a[k], b[k]));
    }
    const int32_t denhalf = denshift ? (1 << (denshift - 1)) : 0;
    const int32_t out = sext(wadd(wadd(del, top), wsub(denhalf, sum) >> denshift), chanshift);
    lms_adapt<NA>(s, b, del, denshift);
    lms_push<NA>(s, out);
    return out;
}

// ---- adaptive Golomb coder, streaming lane-serial form ------------------------------------------
// dyn_comp (codec/ag_enc.c:249-367) consumes residuals one at a time here, so it can be fused
// behind the predictor without storing them.  State carried between symbols: the mean tracker
// mb, zmode, and an open zero run.
struct Golomb {
    uint32_t mb, zmode, inrun, nz, bits;
    uint32_t pb, kb, wb;
    // writer: MSB-first bit accumulator flushed as 32-bit words (first bit = bit 31 of word 0)
    uint64_t acc;
    uint32_t nacc, widx, wcap;
    uint32_t *wp;
};

__device__ __forceinline__ void gol_reset(Golomb &g, uint32_t mb0, uint32_t pb, uint32_t kb)
{
    g.mb = mb0;
    g.zmode = 0;
    g.inrun = 0;
    g.nz = 0;
    g.bits = 0;
    g.pb = pb;
    g.kb = kb;
    g.wb = (1u << kb) - 1;
    g.acc = 0;
    g.nacc = 0;
    g.widx = 0;
}

template <bool WRITE>
__device__ __forceinline__ void gol_put(Golomb &g, uint32_t value, uint32_t nbits)
{
    g.bits += nbits;
    if constexpr (WRITE) {
        const uint64_t v = nbits >= 32 ? (uint64_t)value : (uint64_t)(value & ((1u << nbits) - 1));
        g.acc = (g.acc << nbits) | v;
        g.nacc += nbits;
        if (g.nacc >= 32) {
            g.nacc -= 32;
            if (g.widx < g.wcap) g.wp[g.widx] = (uint32_t)(g.acc >> g.nacc);
            g.widx++;
        }
    }
}

template <bool WRITE>
__device__ __forceinline__ void gol_flush(Golomb &g)
{
    if constexpr (WRITE) {
        if (g.nacc > 0) {
            if (g.widx < g.wcap) g.wp[g.widx] = (uint32_t)(g.acc << (32 - g.nacc));
            g.widx++;
            g.nacc = 0;
        }
    }
}

// dyn_code (codec/ag_enc.c:115-148) of the run length, then mb = 0 (:351-358)
template <bool WRITE>
__device__ __forceinline__ void gol_close_run(Golomb &g)
{
    const uint32_t k = (uint32_t)(lead(g.mb) - 24 + (int32_t)((g.mb + 16u) >> 6));
    const uint32_t mz = ((1u << k) - 1) & g.wb;
    const uint32_t nz = g.nz;
    const uint32_t div = nz / mz;
    uint32_t numBits, value;
    if (div >= kMaxPrefix) {
        numBits = kMaxPrefix + kMaxRunBits;
        value = (((1u << kMaxPrefix) - 1) << kMaxRunBits) + nz;
    } else {
        const uint32_t mod = nz - div * mz;
        const uint32_t de = (mod == 0);
        numBits = div + k + 1 - de;
        value = (((1u << div) - 1) << (numBits - div)) + mod + 1 - de;
        if (numBits > kMaxPrefix + kMaxRunBits) {
            numBits = kMaxPrefix + kMaxRunBits;
            value = (((1u << kMaxPrefix) - 1) << kMaxRunBits) + nz;
        }
    }
    gol_put<WRITE>(g, value, numBits);
    g.mb = 0;
    g.inrun = 0;
}

// One residual.  `last` = this is the final sample of the block (c == numSamples afterwards).
template <bool WRITE>
__device__ __forceinline__ void gol_sym(Golomb &g, int32_t del, bool last, uint32_t bitSize)
{
    if (g.inrun) {
        if (del == 0) {  // :333-349
            g.nz++;
            if (g.nz >= 65535) {
                gol_close_run<WRITE>(g);
                g.zmode = 0;
            } else if (last) {
                gol_close_run<WRITE>(g);
            }
            return;
        }
        gol_close_run<WRITE>(g);  // run ended by a non-zero sample; zmode stays 1
    }
    // :285-309 dyn_code_32bit
    uint32_t m = g.mb >> kQBShift;
    uint32_t k = (uint32_t)lg3a(m);
    k = k > g.kb ? g.kb : k;
    m = (1u << k) - 1;
    const uint32_t a = (uint32_t)(del < 0 ? -del : del);
    const uint32_t n = (a << 1) - ((uint32_t)del >> 31) - g.zmode;
    const uint32_t div = n / m;
    bool esc = true;
    if (div < kMaxPrefix) {
        const uint32_t mod = n - m * div;
        const uint32_t de = (mod == 0);
        const uint32_t numBits = div + k + 1 - de;
        if (numBits <= 25) {
            gol_put<WRITE>(g, (((1u << div) - 1) << (numBits - div)) + mod + 1 - de, numBits);
            esc = false;
        }
    }
    if (esc) {
        gol_put<WRITE>(g, (1u << kMaxPrefix) - 1, kMaxPrefix);
        gol_put<WRITE>(g, n, bitSize);
    }
    // :318-324
    g.mb = g.pb * (n + g.zmode) + g.mb - ((g.pb * g.mb) >> kQBShift);
    if (n > kMeanClamp) g.mb = kMeanClamp;
    g.zmode = 0;
    // :328-331
    if (((g.mb << 2) < (1u << kQBShift)) && !last) {
        g.zmode = 1;
        g.inrun = 1;
        g.nz = 0;
    }
}

// ---- MSB-first bit reader over a byte stream (decoder) ---------------------------------------
// 32 bits starting at absolute bit position `pos` of `base` (bytes past `limitBytes` read as 0).
__device__ __forceinline__ uint32_t peek32(const uint8_t *base, uint64_t limitBytes, uint64_t pos)
{
    const uint64_t byte = pos >> 3;
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint64_t bi = byte + i;
        v = (v << 8) | (bi < limitBytes ? (uint64_t)base[bi] : 0ull);
    }
    return (uint32_t)(v >> (8 - (pos & 7)));
}

__device__ __forceinline__ uint32_t read_bits(const uint8_t *base, uint64_t limitBytes,
                                              uint64_t &pos, uint32_t n)
{
    const uint32_t v = n ? (peek32(base, limitBytes, pos) >> (32 - n)) : 0u;
    pos += n;
    return v;
}

}  // namespace alacdev
