// alac_lms.hpp — tap-parallel adaptive predictor for gfx950 (wave64).
//
// One wavefront walks 8 predictor chains at once: lane = (slot g = lane >> 3, tap k = lane & 7).
// Per step the 8 taps of a chain form their products in parallel and the sum is reduced with a
// three-stage DPP butterfly (quad_perm xor 1, xor 2, row_half_mirror) — no LDS traffic for the
// reduction.  Only the coefficient recurrence is serial; everything that depends on the input data
// alone (tap differences b_k, the early-exit thresholds of the sign-LMS update) is computed off the
// dependent chain.
//
// Reference recurrence: pc_block, codec/dp_enc.c:77-388 (4-tap :116-195, 8-tap :196-340, general
// :341-387 are the same recurrence).  The update loop "walk taps k = na-1 .. 0 while del0 keeps its
// sign" (:143-188) is evaluated without the walk: with t_k = |b_k| >> 9 (del > 0) resp.
// (|b_k| + 511) >> 9 (del < 0) [= -((-|b_k|) >> 9)], tap k is touched iff
// |del| > S_k := sum_{i > k} (na - i) * t_i, because the partial sums only grow.  S_k for both signs
// is an exclusive suffix scan over the 8 lanes of the group (masked DPP butterfly, 7 instructions).
#pragma once

#include "alac_dev.hpp"

namespace alacdev {

constexpr int kDppXor1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // row_half_mirror: lane i <-> 7 - i inside each 8 lanes

template <int CTRL, int RM = 0xf, int BM = 0xf>
__device__ __forceinline__ int32_t dpp_z(int32_t src)
{
    return __builtin_amdgcn_update_dpp(0, src, CTRL, RM, BM, true);
}

__device__ __forceinline__ int32_t med3_i32(int32_t x, int32_t lo, int32_t hi)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}

// all-reduce add over the 8 lanes of a group
__device__ __forceinline__ int32_t group_sum8(int32_t v)
{
    v += dpp_z<kDppXor1>(v);
    v += dpp_z<kDppXor2>(v);
    v += dpp_z<kDppHalfMirror>(v);
    return v;
}

// exclusive suffix sum over the 8 lanes of a group: out_k = sum_{i > k} w_i
//   modd = ~0 on odd taps, mk1 = ~0 on taps with bit 1 set (per-lane constants)
__device__ __forceinline__ uint32_t group_suffix_excl8(uint32_t w, uint32_t modd, uint32_t mk1)
{
    const uint32_t pair = w + (uint32_t)dpp_z<kDppXor1>((int32_t)w);
    const uint32_t quad = pair + (uint32_t)dpp_z<kDppXor2>((int32_t)pair);
    uint32_t s = (uint32_t)dpp_z<kDppXor1>((int32_t)(w & modd));       // even taps: w_{k+1}
    s += (uint32_t)dpp_z<kDppXor2>((int32_t)(pair & mk1));              // taps 0,1 / 4,5: upper pair of the quad
    s += (uint32_t)dpp_z<kDppHalfMirror, 0xf, 0x5>((int32_t)quad);      // taps 0..3: the upper quad
    return s;
}

// per-lane constants of a chain slot
struct LmsLane {
    int32_t k;        // tap index 0..7
    int32_t na;       // taps of this chain (4 or 8); lanes k >= na are inert (their b is forced to 0)
    int32_t wk;       // na - k for k < na, else 0
    int32_t c255;     // 255 on tap 0: folds "denhalf - sum" into the product (see lms8_step)
    uint32_t modd, mk1;
    int32_t jlo, jhi; // coefficients adapt for jlo <= j < jhi  (jlo = na + 1, jhi = pc_block's num)
};

__device__ __forceinline__ LmsLane make_lane(int lane, int na, int num)
{
    LmsLane L;
    L.k = lane & 7;
    L.na = na;
    L.wk = L.k < na ? na - L.k : 0;
    L.c255 = L.k == 0 ? 255 : 0;
    L.modd = (L.k & 1) ? ~0u : 0u;
    L.mk1 = (L.k & 2) ? ~0u : 0u;
    L.jlo = na + 1;
    L.jhi = num;
    return L;
}

// One predictor step of all 8 chains of the wave.
//   a    this lane's coefficient (int16 value kept in an int32)
//   xk   in[j-1-k]  (taps k >= na are handed in[j-1-na] so that b = 0)
//   tp   in[j-1-na] ("top"),  p = in[j] - top (same for the 8 taps: computed once per sample at staging)
// Returns the residual (identical in the 8 lanes of a group).
//   sum1 = (denhalf - sum) >> 9 and del = in[j] - top - sum1 (dp_enc.c:136-139) are folded into
//   del = (in[j] - top) + ((sum + 255) >> 9): -floor((256 - s)/512) == floor((s + 255)/512).
// sign(x) as clamp(x, -1, 1) in one v_med3_i32 with inline constants
__device__ __forceinline__ int32_t sign3(int32_t x)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(x));
    return r;
}

template <bool WIDE, bool MASKED>
__device__ __forceinline__ int32_t lms8_step(int32_t &a, int32_t xk, int32_t tp, int32_t p, int32_t liveMask,
                                             const LmsLane &L, uint32_t chanbits)
{
    // ---- data-only part (off the dependent chain) ----
    const int32_t b = tp - xk;
    const int32_t ab = max(b, -b);
    const uint32_t tpos = (uint32_t)ab >> kDenShift;
    const uint32_t tneg = (uint32_t)(ab + ((1 << kDenShift) - 1)) >> kDenShift;
    int32_t hi, lo;
    if constexpr (!WIDE) {
        // chanbits <= 17: |b| < 2^17 so t <= 256 and every suffix sum < 2^14: both signs share one scan
        const uint32_t w = __umul24(tpos | (tneg << 14), (uint32_t)L.wk);
        const uint32_t s = group_suffix_excl8(w, L.modd, L.mk1);
        hi = (int32_t)(s & 0x3fffu);
        lo = -(int32_t)(s >> 14);
    } else {
        hi = (int32_t)group_suffix_excl8(__umul24(tpos, (uint32_t)L.wk), L.modd, L.mk1);
        lo = -(int32_t)group_suffix_excl8(__umul24(tneg, (uint32_t)L.wk), L.modd, L.mk1);
    }
    int32_t sb = sign3(b);
    if constexpr (MASKED) sb &= liveMask;  // steps outside [jlo, jhi) leave the coefficients alone

    // ---- dependent chain ----
    const int32_t s = group_sum8(__mul24((int32_t)(int16_t)a, b) + L.c255);
    const int32_t del = __builtin_amdgcn_sbfe(p + (s >> kDenShift), 0, chanbits);
    const int32_t e = med3_i32(del, lo, hi) - del;  // < 0: del > S+ (positive side), > 0: del < -S-
    a = __mul24(sign3(e), sb) + a;                  // a -= sign(del) * sign(b) on the touched taps
    return del;
}

}  // namespace alacdev
