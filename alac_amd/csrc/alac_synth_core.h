/*
 * alac_synth_core.h — the deterministic integer-only synthetic PCM generator, ONE
 * source for the host build (alac_synth.c, gcc) and the device build (alac_synth.hip: one thread per frame).
 *
 * Frame f is seeded with 0xA1AC0000 + f and belongs to class f mod 8:
 *   0 silence | 1 full-scale white noise (forces escape) | 2 white noise at -36 dB |
 *   3 AR(2) coloured noise, L/R independent | 4 AR(2) with R = L + small noise (mixRes != 0) |
 *   5 two integer-recurrence sines + dither | 6 sparse full-scale impulses on silence |
 *   7 clipped ramp + DC offset.
 * No libm, no floating point: the same bytes on every host and on the GPU.
 */
#ifndef ALAC_SYNTH_CORE_H
#define ALAC_SYNTH_CORE_H
#ifndef ALAC_SYNTH_FN
#define ALAC_SYNTH_FN static inline
#endif
#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t s; } synth_rng;

ALAC_SYNTH_FN uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* xorshift64*, upper 32 bits */
ALAC_SYNTH_FN uint32_t rnd32(synth_rng *r)
{
    uint64_t x = r->s;
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    r->s = x;
    return (uint32_t)((x * 0x2545F4914F6CDD1Dull) >> 32);
}

ALAC_SYNTH_FN int32_t clip16(int32_t x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }

ALAC_SYNTH_FN void store_le(uint8_t *p, uint32_t bitDepth, int32_t v16, uint32_t low)
{
    switch (bitDepth) {
    case 16:
        p[0] = (uint8_t)v16;
        p[1] = (uint8_t)(v16 >> 8);
        break;
    case 20: { /* 20 significant bits, left-justified in 3 bytes */
        uint32_t v = (((uint32_t)v16 << 4) | (low & 0xf)) << 4;
        p[0] = (uint8_t)v;
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)(v >> 16);
        break;
    }
    case 24: {
        uint32_t v = ((uint32_t)v16 << 8) | (low & 0xff);
        p[0] = (uint8_t)v;
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)(v >> 16);
        break;
    }
    default: {
        uint32_t v = ((uint32_t)v16 << 16) | (low & 0xffff);
        p[0] = (uint8_t)v;
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)(v >> 16);
        p[3] = (uint8_t)(v >> 24);
        break;
    }
    }
}

/* one frame: numSamples sample-frames of `channels` channels into out (packed LE interleaved) */
ALAC_SYNTH_FN void alac_synth_frame_core(uint64_t frameIndex, uint32_t numSamples, uint32_t bitDepth, uint32_t channels,
                      uint8_t *out)
{
    synth_rng r;
    r.s = splitmix64(0xA1AC0000ull + frameIndex) | 1ull;
    const uint32_t cls = (uint32_t)(frameIndex & 7);
    const uint32_t bps = bitDepth == 16 ? 2 : (bitDepth == 32 ? 4 : 3);
    int32_t yl1 = 0, yl2 = 0, yr1 = 0, yr2 = 0;  /* AR(2) state */
    int32_t s1a = 1200, s1b = 0, s2a = 1500, s2b = 0; /* sine recurrences */
    const int32_t amp = 1 + (int32_t)((frameIndex >> 3) & 3); /* per-frame level variety */

    for (uint32_t n = 0; n < numSamples; n++) {
        int32_t l = 0, rr = 0;
        uint32_t lowl = 0, lowr = 0;
        int quiet_low = 0;
        switch (cls) {
        case 0:
            quiet_low = 1;
            break;
        case 1:
            l = (int16_t)rnd32(&r);
            rr = (int16_t)rnd32(&r);
            break;
        case 2:
            l = (int32_t)(rnd32(&r) & 1023) - 512;
            rr = (int32_t)(rnd32(&r) & 1023) - 512;
            break;
        case 3:
        case 4: {
            int32_t e = ((int32_t)(rnd32(&r) & 511) - 256) * amp;
            int32_t y = ((29491 * yl1 - 14746 * yl2) >> 14) + e;
            y = clip16(y);
            yl2 = yl1;
            yl1 = y;
            l = y;
            if (cls == 3) {
                e = ((int32_t)(rnd32(&r) & 511) - 256) * amp;
                y = ((29491 * yr1 - 14746 * yr2) >> 14) + e;
                y = clip16(y);
                yr2 = yr1;
                yr1 = y;
                rr = y;
            } else {
                rr = clip16(l + (int32_t)(rnd32(&r) & 15) - 8);
            }
            break;
        }
        case 5: {
            int32_t a = ((32610 * s1a) >> 14) - s1b;
            a = clip16(a);
            s1b = s1a;
            s1a = a;
            int32_t b = ((31000 * s2a) >> 14) - s2b;
            b = clip16(b);
            s2b = s2a;
            s2a = b;
            uint32_t d = rnd32(&r);
            l = clip16(a * amp / 2 + b / 2 + (int32_t)(d & 3) - 2);
            rr = clip16(a * amp / 2 - b / 2 + (int32_t)((d >> 8) & 3) - 2);
            break;
        }
        case 6: {
            uint32_t d = rnd32(&r);
            if ((d & 511) == 0) l = (d & 0x10000) ? 32767 : -32768;
            if (((d >> 9) & 511) == 0) rr = (d & 0x20000) ? 32767 : -32768;
            quiet_low = 1;
            break;
        }
        default:
            l = clip16(-40000 + 20 * (int32_t)n * amp);
            rr = 12345 + (((n >> 6) & 1) ? 3 : -3);
            break;
        }
        if (bitDepth != 16 && !quiet_low) {
            uint32_t d = rnd32(&r);
            lowl = d & 0xffff;
            lowr = d >> 16;
        }
        if (channels == 2) {
            store_le(out + (size_t)(2 * n) * bps, bitDepth, l, lowl);
            store_le(out + (size_t)(2 * n + 1) * bps, bitDepth, rr, lowr);
        } else {
            store_le(out + (size_t)n * bps, bitDepth, l, lowl);
        }
    }
}

#endif
