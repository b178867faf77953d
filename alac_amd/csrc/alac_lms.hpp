// alac_lms.hpp — the adaptive predictor (pc_block, codec/dp_enc.c:77-388) for gfx950, wave64.
//
// Lane mapping: a chain (one channel of one packet) owns LPC adjacent lanes, each lane holds 4 taps
// (coefficients and the matching 4-sample history window in registers):
//   LPC = 2   8-tap rows: lane h = 0 holds taps 0..3, lane h = 1 taps 4..7; the tap sum of a step is the
//             in-lane 4-term multiply-add chain plus ONE cross-lane add (DPP quad_perm xor 1)
//   LPC = 1   4-tap rows: the whole chain lives in one lane, no cross-lane traffic at all
// so a wave walks 32 (LPC 2) or 64 (LPC 1) chains at once.  With 10 000 stereo packets that is 625 waves,
// i.e. at most one wave per SIMD on 1024 SIMDs: the cost of a step is this wave's own instruction count,
// which is why the taps are folded into the lanes instead of being spread over 8 lanes each.
//
// The early-exit update walk of the reference ("for k = na-1 .. 0 while del0 keeps its sign",
// dp_enc.c:143-188) is evaluated without the walk: with t_i = |b_i| >> 9 (del > 0) resp.
// (|b_i| + 511) >> 9 (del < 0) [= -((-|b_i|) >> 9)], tap k is touched iff
// |del| > S_k := sum_{i > k} (na - i) * t_i, because the partial sums only grow.
#pragma once

#include "alac_dev.hpp"

namespace alacdev {

constexpr int kDppXor1 = 0xB1;  // quad_perm:[1,0,3,2]: partner lane of a 2-lane chain

__device__ __forceinline__ int32_t dpp_xor1(int32_t src)
{
    return __builtin_amdgcn_update_dpp(0, src, kDppXor1, 0xf, 0xf, true);
}

__device__ __forceinline__ int32_t med3_i32(int32_t x, int32_t lo, int32_t hi)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}

// sign(x) as clamp(x, -1, 1) in one v_med3_i32 with inline constants
__device__ __forceinline__ int32_t sign3(int32_t x)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(x));
    return r;
}

// ---- lane mappings --------------------------------------------------------------------------------------
// A chain owns L adjacent lanes (L = 1, 2 or 4, inside one DPP quad), each lane T taps: <T, L> covers T * L >= na taps.
//   <4, 1>  4-tap rows, one lane per chain (64 chains per wave)
//   <4, 2>  8-tap rows, two lanes per chain (32 chains per wave): the shape of the searches at every batch size and
//           of 4- and 8-tap rows in the final pass of mid-size batches
//   <8, 1>  8-tap rows in ONE lane (64 chains per wave): fewest instructions per chain step (throughput regime)
//   <2, 2>  4-tap rows over two lanes, <2, 4> 8-tap rows over four lanes: fewest instructions per WAVE step (the
//           latency regime of the final pass, where the chains of one class fit one wave per SIMD)
// Tap k = T * h + i lives in lane h = lane % L, slot i.

constexpr int kDppXor2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int kDppShl1 = 0xF9;  // quad_perm:[1,2,3,3]: lane q reads lane q + 1 of its quad
constexpr int kDppShl2 = 0xFE;  // quad_perm:[2,3,3,3]: lane q reads lane q + 2 of its quad

template <int CTRL>
__device__ __forceinline__ int32_t dpp_quad(int32_t src)
{
    return __builtin_amdgcn_update_dpp(0, src, CTRL, 0xf, 0xf, true);
}

// per-lane constants of a chain
template <int T>
struct LmsLaneT {
    int32_t h;         // which group of T taps this lane holds (always 0 for L = 1)
    int32_t na;        // taps of the chain (4 or 8)
    int32_t wg[T];     // update weights na - (T h + i), 0 for taps >= na
    int32_t act[T];    // ~0 for taps < na (only consulted where a LANE can hold both kinds: T = 8 with a 4-tap row)
    int32_t c255;      // 255 on the lane holding tap 0: folds "denhalf - sum" into the product chain
    int32_t carry1;    // ~0 where lane h + 1 belongs to the chain: it takes that lane's threshold total
    int32_t carry2;    // ~0 where lane h + 2 belongs to the chain (L = 4)
    int32_t jlo, jhi;  // coefficients adapt for jlo <= j < jhi (jlo = na + 1, jhi = pc_block's num)
};

template <int T, int L>
__device__ __forceinline__ LmsLaneT<T> make_lane(int lane, int na, int num)
{
    LmsLaneT<T> R;
    R.h = L == 1 ? 0 : (lane & (L - 1));
    R.na = na;
#pragma unroll
    for (int i = 0; i < T; i++) {
        const int kk = T * R.h + i;
        R.wg[i] = kk < na ? na - kk : 0;
        R.act[i] = kk < na ? -1 : 0;
    }
    R.c255 = R.h == 0 ? 255 : 0;
    R.carry1 = (R.h + 1 < L) ? -1 : 0;
    R.carry2 = (R.h + 2 < L) ? -1 : 0;
    R.jlo = na + 1;
    R.jhi = num;
    return R;
}

// One predictor step of every chain of the wave.
//   a[i]  coefficient of tap T h + i (int16 value carried in an int32; the wrap is applied where it is read)
//   w[i]  in[j-1-(T h + i)]   (lanes holding no active tap are fed zeros: b = 0, nothing happens)
//   tp    in[j-1-na] ("top"),  cu = in[j]
// Returns the residual.  sum1 = (denhalf - sum) >> 9 and del = in[j] - top - sum1 (dp_enc.c:136-139)
// are folded into del = (in[j] - top) + ((sum + 255) >> 9): -floor((256 - s)/512) == floor((s + 255)/512).
// DEAD: the lane may hold taps beyond its row's own count (a 4-tap row in an 8-tap lane: k_class_final<.., 8>); kernels whose
// rows always fill the lane (the in-lane searches) pass false and save the mask per tap
template <int T, int L, bool MASKED, bool DEAD = (T > 4)>
__device__ __forceinline__ int32_t lms_step(int32_t (&a)[T], const int32_t (&w)[T], int32_t tp, int32_t cu,
                                            int32_t liveMask, const LmsLaneT<T> &R, uint32_t chanbits)
{
    // A wave that has a SIMD to itself pays for every instruction it issues (a wave64 integer instruction occupies
    // the SIMD for ~4.3 cycles, tools/op_rate_microbench.hip), so the step is written for the fewest instructions,
    // not for the shortest dependent chain: the residual first, then ONE set of thresholds with the rounding its
    // sign selects (an earlier version evaluated both roundings ahead of del: 73 instructions per step against 50).
    int32_t b[T];
#pragma unroll
    for (int i = 0; i < T; i++) b[i] = tp - w[i];
    const int32_t p = cu - tp;
    int32_t s = R.c255;
#pragma unroll
    for (int i = 0; i < T; i++) s = __mul24((int32_t)(int16_t)a[i], b[i]) + s;
    if constexpr (L >= 2) s += dpp_xor1(s);
    if constexpr (L == 4) s += dpp_quad<kDppXor2>(s);
    const int32_t del = __builtin_amdgcn_sbfe(p + (s >> kDenShift), 0, chanbits);

    // coefficient walk: t_i = (|b_i| + rc) >> 9 with rc = 511 for del < 0 (the arithmetic shift of a negative
    // product rounds away from zero, dp_enc.c:176), tap k is touched iff |del| > S_k
    const int32_t nd = -del;
    const int32_t adel = max(del, nd);
    int32_t nsg = sign3(nd);                // -sign(del): what a touched tap adds per sign(b)
    if constexpr (MASKED) nsg &= liveMask;  // steps outside [jlo, jhi) leave the coefficients alone
    const int32_t rc = (del >> 31) & ((1 << kDenShift) - 1);
    int32_t sb[T];
    uint32_t t[T];
#pragma unroll
    for (int i = 0; i < T; i++) {
        sb[i] = sign3(b[i]);
        // A lane that holds no active tap at all is fed zeros (b = 0: nothing happens).  With 8 taps in one lane a 4-tap
        // row (the narrower channel of a packet of the 8-tap class) has dead taps BESIDE live ones: their window holds
        // real samples, so their sign is forced to 0 — then t = 0, the coefficient stays 0 and the tap sum ignores it.
        if constexpr (DEAD) sb[i] &= R.act[i];
        t[i] = (uint32_t)(__mul24(sb[i], b[i]) + rc) >> kDenShift;  // |b| = sign(b) * b
    }
    int32_t S[T];  // in-lane part of S_k, from the top tap down
    S[T - 1] = 0;
#pragma unroll
    for (int i = T - 1; i > 0; i--) S[i - 1] = (int32_t)__umul24(t[i], (uint32_t)R.wg[i]) + S[i];
    int32_t adj = adel;
    if constexpr (L >= 2) {
        // the lower taps also count everything the lanes above hold: |del| > S + c  <=>  |del| - c > S
        const int32_t tot = (int32_t)__umul24(t[0], (uint32_t)R.wg[0]) + S[0];
        if constexpr (L == 2) {
            adj = adel - (dpp_xor1(tot) & R.carry1);
        } else {
            const int32_t up1 = dpp_quad<kDppShl1>(tot) & R.carry1;          // total of lane h + 1
            const int32_t pair = tot + up1;                                   // lanes h, h + 1
            adj = adel - up1 - (dpp_quad<kDppShl2>(pair) & R.carry2);         // + lanes h + 2, h + 3
        }
    }
#pragma unroll
    for (int i = 0; i < T; i++) a[i] = __mul24(adj > S[i] ? nsg : 0, sb[i]) + a[i];
    return del;
}

// the two shapes every kernel used before the lane mappings became a template parameter
using LmsLane = LmsLaneT<4>;
template <int LPC, bool MASKED>
__device__ __forceinline__ int32_t lms4_step(int32_t (&a)[4], const int32_t (&w)[4], int32_t tp, int32_t cu,
                                             int32_t liveMask, const LmsLane &R, uint32_t chanbits)
{
    return lms_step<4, LPC, MASKED>(a, w, tp, cu, liveMask, R, chanbits);
}

// ---- decode direction (unpc_block, codec/dp_dec.c:55-381) with the same lane mapping ----
// The residual `del` is an INPUT here, so the coefficient update (same thresholds, same signs as pc_block) is off
// the dependent chain altogether; what is serial is out[j-1] -> b_0 -> tap sum -> out[j].  The history windows
// hold the chain's own outputs: lane h = 0 keeps out[j-1..j-4], lane h = 1 out[j-5..j-8]; the sample leaving
// lane 0's window enters lane 1's, and the one leaving the window of the last active tap is the next "top"
// (one DPP exchange per step for both).
struct LmsDecLane {
    int32_t h;
    int32_t wg[4];
    int32_t c255, carryMask;
    int32_t topMine;   // -1 on the lane whose dropped sample becomes the next top (h = 1 for 8 taps, h = 0 for 4)
    int32_t takeOut;   // -1 on h = 0: the new output enters this lane's window
    int32_t feed;      // 0 on a lane that holds no active tap (h = 1 of a 4-tap chain): it is fed zeros, so its
                       // b, its signs and therefore its coefficients stay 0
};

__device__ __forceinline__ LmsDecLane make_dec_lane(int lane, int na)
{
    LmsDecLane L;
    L.h = lane & 1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int kk = 4 * L.h + i;
        L.wg[i] = kk < na ? na - kk : 0;
    }
    L.c255 = L.h == 0 ? 255 : 0;
    L.carryMask = L.h == 0 ? -1 : 0;
    L.topMine = (L.h == (na == 8 ? 1 : 0)) ? -1 : 0;
    L.takeOut = L.h == 0 ? -1 : 0;
    L.feed = (L.h == 0 || na == 8) ? -1 : 0;
    return L;
}

// one unpc step: returns out[j]; updates a[], the window w[] and tp (the "top" of the NEXT step)
__device__ __forceinline__ int32_t lms4_step_dec(int32_t (&a)[4], int32_t (&w)[4], int32_t &tp, int32_t del,
                                                 const LmsDecLane &L, uint32_t chanbits)
{
    int32_t b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) b[i] = tp - w[i];
    // out[j] = sext(del + top + ((denhalf - sum) >> 9)), -((256 - s) >> 9) == (s + 255) >> 9
    int32_t s = L.c255;
#pragma unroll
    for (int i = 0; i < 4; i++) s = __mul24((int32_t)(int16_t)a[i], b[i]) + s;
    s += dpp_xor1(s);
    const int32_t out = __builtin_amdgcn_sbfe(del + tp - (s >> kDenShift), 0, chanbits);

    // coefficient update, driven by del (dp_dec.c:143-188 == the encoder's walk, same form as lms4_step)
    const int32_t nd = -del;
    const int32_t adel = max(del, nd);
    const int32_t nsg = sign3(nd);
    const int32_t rc = (del >> 31) & ((1 << kDenShift) - 1);
    int32_t sb[4];
    uint32_t t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        sb[i] = sign3(b[i]);
        t[i] = (uint32_t)(__mul24(sb[i], b[i]) + rc) >> kDenShift;
    }
    int32_t S[4];
    S[3] = 0;
#pragma unroll
    for (int i = 3; i > 0; i--) S[i - 1] = (int32_t)__umul24(t[i], (uint32_t)L.wg[i]) + S[i];
    const int32_t tot = (int32_t)__umul24(t[0], (uint32_t)L.wg[0]) + S[0];
    const int32_t adj = adel - (dpp_xor1(tot) & L.carryMask);
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = __mul24(adj > S[i] ? nsg : 0, sb[i]) + a[i];
    // windows: lane 0's oldest sample moves to lane 1, the oldest sample of the last active tap is the next top
    const int32_t x = dpp_xor1(w[3]) & L.feed;
    tp = L.topMine ? w[3] : x;
    w[3] = w[2];
    w[2] = w[1];
    w[1] = w[0];
    w[0] = L.takeOut ? out : x;
    return out;
}

}  // namespace alacdev
