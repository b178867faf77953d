/*
 * alac_oracle.c — TEST INFRASTRUCTURE ONLY (see alac_oracle.h).
 *
 * Plain-C restatement of the reference's ALAC encode/decode hot path, written from the
 * algorithm's definition (one generic predictor loop instead of the reference's unrolled
 * 4/8-tap copies, a bit-position cursor instead of BitBuffer).  Compile with -fwrapv.
 * Each function cites the reference lines it restates (relative to /root/reference).
 */
#include "alac_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * bit I/O — codec/ALACBitUtilities.c
 * ---------------------------------------------------------------------------------------- */

/* BitBufferWrite, ALACBitUtilities.c:212-249: MSB-first, overwrites exactly numBits bits */
void oalac_put_bits(uint8_t *buf, uint64_t *bitpos, uint32_t value, uint32_t numBits)
{
    uint64_t pos = *bitpos;
    while (numBits > 0) {
        uint32_t room = 8 - (uint32_t)(pos & 7);
        uint32_t take = numBits < room ? numBits : room;
        uint32_t shift = room - take;
        uint8_t mask = (uint8_t)((0xffu >> (8 - take)) << shift);
        uint8_t bits = (uint8_t)(((value >> (numBits - take)) << shift) & mask);
        uint8_t *p = buf + (pos >> 3);
        *p = (uint8_t)((*p & ~mask) | bits);
        numBits -= take;
        pos += take;
    }
    *bitpos = pos;
}

static inline uint32_t byte_at(const uint8_t *buf, uint64_t bufbytes, uint64_t i)
{
    return i < bufbytes ? buf[i] : 0u;
}

/* big-endian 32-bit fetch at a byte offset (ag_dec.c:115-124 read32bit), bounded */
static inline uint32_t be32_at(const uint8_t *buf, uint64_t bufbytes, uint64_t i)
{
    return (byte_at(buf, bufbytes, i) << 24) | (byte_at(buf, bufbytes, i + 1) << 16) |
           (byte_at(buf, bufbytes, i + 2) << 8) | byte_at(buf, bufbytes, i + 3);
}

/* BitBufferRead / ReadSmall / ReadOne, ALACBitUtilities.c:42-107 (numBits <= 16 there; this
 * generalisation to <= 32 returns the same bits) */
uint32_t oalac_get_bits(const uint8_t *buf, uint64_t bufbytes, uint64_t *bitpos, uint32_t numBits)
{
    uint64_t pos = *bitpos;
    uint32_t out = 0;
    uint32_t n = numBits;
    while (n > 0) {
        uint32_t room = 8 - (uint32_t)(pos & 7);
        uint32_t take = n < room ? n : room;
        uint32_t b = byte_at(buf, bufbytes, pos >> 3);
        b = (b >> (room - take)) & (0xffu >> (8 - take));
        out = (out << take) | b;
        n -= take;
        pos += take;
    }
    *bitpos = pos;
    return out;
}

/* ------------------------------------------------------------------------------------------
 * predictor — codec/dp_enc.c, codec/dp_dec.c
 * ---------------------------------------------------------------------------------------- */

/* dp_enc.c:49-60 (AINIT 38, BINIT -29, CINIT -2 from dplib.h:43-47) */
void oalac_init_coefs(int16_t *coefs, uint32_t denshift, int32_t numPairs)
{
    int32_t den = 1 << denshift;
    coefs[0] = (int16_t)((38 * den) >> 4);
    coefs[1] = (int16_t)((-29 * den) >> 4);
    coefs[2] = (int16_t)((-2 * den) >> 4);
    for (int32_t k = 3; k < numPairs; k++) coefs[k] = 0;
}

/* dp_enc.c:69-75 */
static inline int32_t sign_of(int32_t i) { return (i > 0) - (i < 0); }

static inline int32_t sext(int32_t x, uint32_t chanshift)
{
    return (int32_t)((uint32_t)x << chanshift) >> chanshift;
}

/*
 * pc_block, dp_enc.c:77-388.  The reference's numactive==4 (:116-195) and ==8 (:196-340)
 * branches are unrolled copies of the general loop (:341-387): same sum in wrapping int32,
 * same int16 coefficient storage, and the last tap (k = 0) is updated without the del0 step,
 * which nothing reads afterwards.  One loop therefore restates all three.
 */
void oalac_pc_block(int32_t *in, int32_t *pc, int32_t num, int16_t *coefs, int32_t numactive,
                    uint32_t chanbits, uint32_t denshift)
{
    uint32_t chanshift = 32 - chanbits;
    int32_t denhalf = denshift ? (1 << (denshift - 1)) : 0;

    pc[0] = in[0];
    if (numactive == 0) { /* :91-97 */
        if (num > 1 && in != pc) memcpy(&pc[1], &in[1], (size_t)(num - 1) * sizeof(int32_t));
        return;
    }
    if (numactive == 31) { /* :98-107 first-order delta */
        for (int32_t j = 1; j < num; j++) pc[j] = sext(in[j] - in[j - 1], chanshift);
        return;
    }
    /* :108-112 warm-up: runs to numactive even when num is smaller (dplib.h:52) */
    for (int32_t j = 1; j <= numactive; j++) pc[j] = sext(in[j] - in[j - 1], chanshift);

    for (int32_t j = numactive + 1; j < num; j++) {
        int32_t top = in[j - numactive - 1];
        const int32_t *pin = in + j - 1;
        int32_t sum = 0;
        for (int32_t k = 0; k < numactive; k++) sum -= coefs[k] * (top - pin[-k]);
        int32_t del = sext(in[j] - top - ((sum + denhalf) >> denshift), chanshift);
        pc[j] = del;
        int32_t del0 = del;
        int32_t sg = sign_of(del);
        if (sg > 0) {
            for (int32_t k = numactive - 1; k >= 0; k--) {
                int32_t dd = top - pin[-k];
                int32_t sgn = sign_of(dd);
                coefs[k] = (int16_t)(coefs[k] - sgn);
                del0 -= (numactive - k) * ((sgn * dd) >> denshift);
                if (del0 <= 0) break;
            }
        } else if (sg < 0) {
            for (int32_t k = numactive - 1; k >= 0; k--) {
                int32_t dd = top - pin[-k];
                int32_t sgn = sign_of(dd);
                coefs[k] = (int16_t)(coefs[k] + sgn);
                del0 -= (numactive - k) * ((-sgn * dd) >> denshift);
                if (del0 >= 0) break;
            }
        }
    }
}

/* unpc_block, dp_dec.c:55-381 (same remark on the 4/8-tap copies, :105-334 vs :335-380) */
void oalac_unpc_block(int32_t *pc, int32_t *out, int32_t num, int16_t *coefs, int32_t numactive,
                      uint32_t chanbits, uint32_t denshift)
{
    uint32_t chanshift = 32 - chanbits;
    int32_t denhalf = denshift ? (1 << (denshift - 1)) : 0;

    out[0] = pc[0];
    if (numactive == 0) {
        if (num > 1 && pc != out) memcpy(&out[1], &pc[1], (size_t)(num - 1) * sizeof(int32_t));
        return;
    }
    if (numactive == 31) { /* :74-95, in-place safe */
        int32_t prev = out[0];
        for (int32_t j = 1; j < num; j++) {
            prev = sext(pc[j] + prev, chanshift);
            out[j] = prev;
        }
        return;
    }
    for (int32_t j = 1; j <= numactive; j++) out[j] = sext(pc[j] + out[j - 1], chanshift);

    for (int32_t j = numactive + 1; j < num; j++) {
        int32_t top = out[j - numactive - 1];
        const int32_t *pout = out + j - 1;
        int32_t sum = 0;
        for (int32_t k = 0; k < numactive; k++) sum += coefs[k] * (pout[-k] - top);
        int32_t del = pc[j];
        int32_t del0 = del;
        int32_t sg = sign_of(del);
        out[j] = sext(del + top + ((sum + denhalf) >> denshift), chanshift);
        if (sg > 0) {
            for (int32_t k = numactive - 1; k >= 0; k--) {
                int32_t dd = top - pout[-k];
                int32_t sgn = sign_of(dd);
                coefs[k] = (int16_t)(coefs[k] - sgn);
                del0 -= (numactive - k) * ((sgn * dd) >> denshift);
                if (del0 <= 0) break;
            }
        } else if (sg < 0) {
            for (int32_t k = numactive - 1; k >= 0; k--) {
                int32_t dd = top - pout[-k];
                int32_t sgn = sign_of(dd);
                coefs[k] = (int16_t)(coefs[k] + sgn);
                del0 -= (numactive - k) * ((-sgn * dd) >> denshift);
                if (del0 >= 0) break;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * adaptive Golomb coder — codec/ag_enc.c, codec/ag_dec.c
 * ---------------------------------------------------------------------------------------- */

/* ag_enc.c:65-77 / ag_dec.c:88-100: count of leading zero bits, lead(0) == 32 */
static inline int32_t lead(uint32_t m) { return m ? __builtin_clz(m) : 32; }
/* ag_enc.c:81-89 */
static inline int32_t lg3a(int32_t x) { return 31 - lead((uint32_t)(x + 3)); }

#define MAX_PREFIX 9        /* aglib.h:50-52 MAX_PREFIX_16 == MAX_PREFIX_32 */
#define MAX_DATATYPE_BITS_16 16
#define MMULSHIFT 2         /* aglib.h:42 */
#define MDENSHIFT (OALAC_QBSHIFT - MMULSHIFT - 1)
#define MOFF (1 << (MDENSHIFT - 2))
#define BITOFF 24
#define N_MAX_MEAN_CLAMP 0xffffu /* ag_enc.c:47-48 */

/* dyn_comp, ag_enc.c:249-367 with dyn_code_32bit :151-184, dyn_code :115-148 and the two
 * dyn_jam writers :187-247 (expressed through oalac_put_bits: same bits, MSB first).
 * rowJump (:264,:312-316) is fw - sw == 0 at every reference call site, so it is omitted. */
int32_t oalac_dyn_comp(uint32_t mb0, uint32_t pb, uint32_t kb, int32_t *pc, uint8_t *buf,
                       uint64_t *bitpos, int32_t numSamples, int32_t bitSize, uint32_t *outNumBits)
{
    *outNumBits = 0;
    if (bitSize < 1 || bitSize > 32) return OALAC_ParamError;

    uint32_t wb = (1u << kb) - 1; /* ag_dec.c:78 */
    uint64_t pos = *bitpos, start = *bitpos;
    uint32_t mb = mb0;
    int32_t zmode = 0;
    int32_t c = 0;

    while (c < numSamples) {
        uint32_t m = mb >> OALAC_QBSHIFT;
        uint32_t k = (uint32_t)lg3a((int32_t)m);
        if (k > kb) k = kb;
        m = (1u << k) - 1;

        int32_t del = pc[c++];
        int32_t a = del < 0 ? -del : del; /* abs_func :91-99 (wraps like the reference on INT_MIN) */
        uint32_t n = ((uint32_t)a << 1) - (uint32_t)((del >> 31) & 1) - (uint32_t)zmode;

        /* dyn_code_32bit */
        uint32_t div = n / m;
        int escaped = 1;
        if (div < MAX_PREFIX) {
            uint32_t mod = n - m * div;
            uint32_t de = (mod == 0);
            uint32_t numBits = div + k + 1 - de;
            uint32_t value = (((1u << div) - 1) << (numBits - div)) + mod + 1 - de;
            if (numBits <= 25) {
                oalac_put_bits(buf, &pos, value, numBits);
                escaped = 0;
            }
        }
        if (escaped) {
            oalac_put_bits(buf, &pos, (1u << MAX_PREFIX) - 1, MAX_PREFIX);
            oalac_put_bits(buf, &pos, n, (uint32_t)bitSize);
        }

        mb = pb * (n + (uint32_t)zmode) + mb - ((pb * mb) >> OALAC_QBSHIFT);
        if (n > N_MAX_MEAN_CLAMP) mb = N_MAX_MEAN_CLAMP;
        zmode = 0;

        if (((mb << MMULSHIFT) < (1u << OALAC_QBSHIFT)) && (c < numSamples)) {
            zmode = 1;
            uint32_t nz = 0;
            while (c < numSamples && pc[c] == 0) {
                ++c;
                ++nz;
                if (nz >= 65535) {
                    zmode = 0;
                    break;
                }
            }
            k = (uint32_t)(lead(mb) - BITOFF + (int32_t)((mb + MOFF) >> MDENSHIFT));
            uint32_t mz = ((1u << k) - 1) & wb;

            /* dyn_code (16-bit) */
            uint32_t d = nz / mz;
            uint32_t numBits, value;
            if (d >= MAX_PREFIX) {
                numBits = MAX_PREFIX + MAX_DATATYPE_BITS_16;
                value = (((1u << MAX_PREFIX) - 1) << MAX_DATATYPE_BITS_16) + nz;
            } else {
                uint32_t mod = nz % mz;
                uint32_t de = (mod == 0);
                numBits = d + k + 1 - de;
                value = (((1u << d) - 1) << (numBits - d)) + mod + 1 - de;
                if (numBits > MAX_PREFIX + MAX_DATATYPE_BITS_16) {
                    numBits = MAX_PREFIX + MAX_DATATYPE_BITS_16;
                    value = (((1u << MAX_PREFIX) - 1) << MAX_DATATYPE_BITS_16) + nz;
                }
            }
            oalac_put_bits(buf, &pos, value, numBits);
            mb = 0;
        }
    }
    *outNumBits = (uint32_t)(pos - start);
    *bitpos = pos;
    return OALAC_noErr;
}

/* getstreambits, ag_dec.c:135-168 */
static uint32_t stream_bits(const uint8_t *buf, uint64_t bufbytes, uint64_t bitoffset, int32_t numbits)
{
    uint64_t pos = bitoffset;
    return oalac_get_bits(buf, bufbytes, &pos, (uint32_t)numbits);
}

/* dyn_decomp, ag_dec.c:272-362 with dyn_get_32bit :220-270 and dyn_get :171-217 */
int32_t oalac_dyn_decomp(uint32_t mb0, uint32_t pb, uint32_t kb, uint8_t *buf, uint64_t bufbytes,
                         uint64_t *bitpos, int32_t *pc, int32_t numSamples, int32_t maxSize,
                         uint32_t *outNumBits)
{
    if (!buf || !pc || !outNumBits) return OALAC_ParamError;
    *outNumBits = 0;
    uint32_t wb = (1u << kb) - 1;
    uint64_t pos = *bitpos, start = *bitpos;
    uint64_t maxPos = bufbytes * 8;
    uint32_t mb = mb0;
    int32_t zmode = 0;
    int32_t c = 0;
    int32_t status = OALAC_noErr;

    while (c < numSamples) {
        if (!(pos < maxPos)) { /* :302 */
            status = OALAC_ParamError;
            break;
        }
        uint32_t m = mb >> OALAC_QBSHIFT;
        uint32_t k = (uint32_t)lg3a((int32_t)m);
        if (k > kb) k = kb;
        m = (1u << k) - 1;

        /* dyn_get_32bit */
        uint32_t n;
        {
            uint32_t streamlong = be32_at(buf, bufbytes, pos >> 3) << (pos & 7);
            uint32_t pre = (uint32_t)lead(~streamlong);
            if (pre >= MAX_PREFIX) {
                n = stream_bits(buf, bufbytes, pos + MAX_PREFIX, maxSize);
                pos += MAX_PREFIX + (uint32_t)maxSize;
            } else {
                pos += pre + 1;
                n = pre;
                if (k != 1) {
                    streamlong <<= pre + 1;
                    uint32_t v = streamlong >> (32 - k);
                    pos += k - 1;
                    n = pre * m;
                    if (v >= 2) {
                        n += v - 1;
                        pos += 1;
                    }
                }
            }
        }

        uint32_t ndecode = n + (uint32_t)zmode;
        int32_t multiplier = -(int32_t)(ndecode & 1);
        multiplier |= 1;
        pc[c++] = (int32_t)((ndecode + 1) >> 1) * multiplier;

        mb = pb * (n + (uint32_t)zmode) + mb - ((pb * mb) >> OALAC_QBSHIFT);
        if (n > N_MAX_MEAN_CLAMP) mb = N_MAX_MEAN_CLAMP;
        zmode = 0;

        if (((mb << MMULSHIFT) < (1u << OALAC_QBSHIFT)) && (c < numSamples)) {
            zmode = 1;
            k = (uint32_t)(lead(mb) - BITOFF + (int32_t)((mb + MOFF) >> MDENSHIFT));
            uint32_t mz = ((1u << k) - 1) & wb;

            /* dyn_get */
            uint32_t streamlong = be32_at(buf, bufbytes, pos >> 3) << (pos & 7);
            uint32_t pre = (uint32_t)lead(~streamlong);
            uint32_t nz;
            if (pre >= MAX_PREFIX) {
                pos += MAX_PREFIX;
                streamlong <<= MAX_PREFIX;
                nz = streamlong >> (32 - MAX_DATATYPE_BITS_16);
                pos += MAX_DATATYPE_BITS_16;
            } else {
                pos += pre + 1;
                streamlong <<= pre + 1;
                uint32_t v = streamlong >> (32 - k);
                pos += k;
                nz = pre * mz + v - 1;
                if (v < 2) {
                    nz -= (v - 1);
                    pos -= 1;
                }
            }
            if (!((uint64_t)c + nz <= (uint64_t)numSamples)) { /* :341 */
                status = OALAC_ParamError;
                break;
            }
            for (uint32_t j = 0; j < nz; j++) pc[c++] = 0;
            if (nz >= 65535) zmode = 0;
            mb = 0;
        }
    }
    *outNumBits = (uint32_t)(pos - start);
    *bitpos = pos;
    if (status == OALAC_noErr && (pos + 7) / 8 > bufbytes) status = OALAC_ParamError; /* :359 */
    return status;
}

/* ------------------------------------------------------------------------------------------
 * stereo mix / un-mix — codec/matrix_enc.cu, codec/ALACDecoder.cu
 * ---------------------------------------------------------------------------------------- */

static inline int32_t load_sample(const uint8_t *p, uint32_t bitDepth)
{
    switch (bitDepth) {
    case 16: return (int16_t)(p[0] | (p[1] << 8));
    case 20: /* left-justified in 3 bytes: (x<<8)>>12, matrix_enc.cu:129-134 */
        return (int32_t)(((uint32_t)p[2] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 8)) >> 12;
    case 24: /* (x<<8)>>8, matrix_enc.cu:197-202 */
        return (int32_t)(((uint32_t)p[2] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 8)) >> 8;
    default:
        return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
                         ((uint32_t)p[3] << 24));
    }
}

static inline uint32_t bytes_per_sample(uint32_t bitDepth) { return bitDepth == 16 ? 2 : bitDepth == 32 ? 4 : 3; }

/* mix16 matrix_enc.cu:72-118, mix20 :120-183, mix24 :186-323, mix32 :330-425:
 * mixres != 0: u = (mixres*l + (2^mixbits - mixres)*r) >> mixbits, v = l - r; mixres == 0: u = l, v = r.
 * 24/32-bit with bytesShifted: low bytes go to shiftUV interleaved, then l,r >>= shift. */
static void mix_strided(const uint8_t *pcm, uint32_t stride, uint32_t bitDepth, int32_t *u, int32_t *v,
                        int32_t numSamples, int32_t mixbits, int32_t mixres, uint16_t *shiftUV, int32_t bytesShifted);

void oalac_mix(const uint8_t *pcm, uint32_t bitDepth, int32_t *u, int32_t *v, int32_t numSamples,
               int32_t mixbits, int32_t mixres, uint16_t *shiftUV, int32_t bytesShifted)
{
    mix_strided(pcm, 2, bitDepth, u, v, numSamples, mixbits, mixres, shiftUV, bytesShifted);
}

/* `stride` = interleaved channels of the input (the mixNN stride argument, matrix_enc.cu:72): the pair is
 * channels 0 and 1 at `pcm` of a stride-channel frame */
static void mix_strided(const uint8_t *pcm, uint32_t stride, uint32_t bitDepth, int32_t *u, int32_t *v,
                        int32_t numSamples, int32_t mixbits, int32_t mixres, uint16_t *shiftUV, int32_t bytesShifted)
{
    uint32_t bps = bytes_per_sample(bitDepth);
    int32_t shift = bytesShifted * 8;
    uint32_t mask = (1u << shift) - 1;
    int32_t m2 = (1 << mixbits) - mixres;
    for (int32_t z = 0; z < numSamples; z++) {
        int32_t l = load_sample(pcm + ((size_t)z * stride) * bps, bitDepth);
        int32_t r = load_sample(pcm + ((size_t)z * stride + 1) * bps, bitDepth);
        if (bytesShifted != 0) {
            shiftUV[2 * z + 0] = (uint16_t)((uint32_t)l & mask);
            shiftUV[2 * z + 1] = (uint16_t)((uint32_t)r & mask);
            l >>= shift;
            r >>= shift;
        }
        if (mixres != 0) {
            u[z] = (mixres * l + m2 * r) >> mixbits;
            v[z] = l - r;
        } else {
            u[z] = l;
            v[z] = r;
        }
    }
}

static inline void store_sample(uint8_t *p, uint32_t bitDepth, int32_t x)
{
    switch (bitDepth) {
    case 16:
        p[0] = (uint8_t)x;
        p[1] = (uint8_t)(x >> 8);
        break;
    case 20: /* ALACDecoder.cu:245-257: value << 4, 3 bytes */
        x = (int32_t)((uint32_t)x << 4);
        /* fallthrough */
    case 24:
        p[0] = (uint8_t)x;
        p[1] = (uint8_t)(x >> 8);
        p[2] = (uint8_t)(x >> 16);
        break;
    default:
        p[0] = (uint8_t)x;
        p[1] = (uint8_t)(x >> 8);
        p[2] = (uint8_t)(x >> 16);
        p[3] = (uint8_t)(x >> 24);
        break;
    }
}

/* gpu_unmix16 ALACDecoder.cu:193-223, unmix20 :227-278, unmix24 :282-338, unmix32 :344-383 */
static void unmix_strided(const int32_t *u, const int32_t *v, uint8_t *pcm, uint32_t stride, uint32_t bitDepth,
                          int32_t numSamples, int32_t mixbits, int32_t mixres, const uint16_t *shiftUV,
                          int32_t bytesShifted);

void oalac_unmix(const int32_t *u, const int32_t *v, uint8_t *pcm, uint32_t bitDepth,
                 int32_t numSamples, int32_t mixbits, int32_t mixres, const uint16_t *shiftUV,
                 int32_t bytesShifted)
{
    unmix_strided(u, v, pcm, 2, bitDepth, numSamples, mixbits, mixres, shiftUV, bytesShifted);
}

static void unmix_strided(const int32_t *u, const int32_t *v, uint8_t *pcm, uint32_t stride, uint32_t bitDepth,
                          int32_t numSamples, int32_t mixbits, int32_t mixres, const uint16_t *shiftUV,
                          int32_t bytesShifted)
{
    uint32_t bps = bytes_per_sample(bitDepth);
    int32_t shift = bytesShifted * 8;
    for (int32_t z = 0; z < numSamples; z++) {
        int32_t l, r;
        if (mixres != 0) {
            l = u[z] + v[z] - ((mixres * v[z]) >> mixbits);
            r = l - v[z];
        } else {
            l = u[z];
            r = v[z];
        }
        if (bytesShifted != 0 && bitDepth >= 24) { /* the 16/20-bit kernels have no shift path */
            l = (int32_t)(((uint32_t)l << shift) | shiftUV[2 * z + 0]);
            r = (int32_t)(((uint32_t)r << shift) | shiftUV[2 * z + 1]);
        }
        store_sample(pcm + ((size_t)z * stride) * bps, bitDepth, l);
        store_sample(pcm + ((size_t)z * stride + 1) * bps, bitDepth, r);
    }
}

/* ------------------------------------------------------------------------------------------
 * encoder driver — codec/ALACEncoder.cu
 * ---------------------------------------------------------------------------------------- */

struct oalac_encoder {
    uint32_t frameSize, bitDepth, numChannels, sampleRate;
    uint32_t bufLen; /* frameSize padded so the warm-up reads of pc_block stay inside */
    int16_t coefsU[OALAC_MAX_CHANNELS][OALAC_MAX_SEARCHES][OALAC_MAX_COEFS];
    int16_t coefsV[OALAC_MAX_CHANNELS][OALAC_MAX_SEARCHES][OALAC_MAX_COEFS];
    int32_t *mixU, *mixV, *predU, *predV;
    uint16_t *shiftUV;
    uint8_t *work;
    uint32_t workBytes;
    uint32_t totalBytes, maxFrameBytes;
    uint32_t info[6];
    oalac_hooks hooks;
    int fastMode; /* SetFastMode, ALACEncoder.h:44 */
};

static const oalac_hooks k_own_hooks = {oalac_pc_block, oalac_unpc_block, oalac_dyn_comp, oalac_dyn_decomp};

uint32_t oalac_max_packet_bytes(uint32_t frameSize, uint32_t bitDepth, uint32_t numChannels)
{
    /* worst case before the "too big" rewind (ALACEncoder.cu:537-543): every symbol escaped
     * (9 + chanBits bits) plus a run code, plus header; generous and > mMaxOutputBytes (:1489) */
    (void)bitDepth;
    return frameSize * numChannels * 8 + 256;
}

/* InitializeEncoder, ALACEncoder.cu:1457-1535 */
oalac_encoder *oalac_encoder_new(uint32_t frameSize, uint32_t bitDepth, uint32_t numChannels,
                                 uint32_t sampleRate)
{
    if (!(bitDepth == 16 || bitDepth == 20 || bitDepth == 24 || bitDepth == 32)) return NULL;
    if (numChannels < 1 || numChannels > OALAC_MAX_CHANNELS || frameSize == 0) return NULL;
    oalac_encoder *e = (oalac_encoder *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->frameSize = frameSize;
    e->bitDepth = bitDepth;
    e->numChannels = numChannels;
    e->sampleRate = sampleRate;
    e->bufLen = frameSize + 64;
    e->mixU = (int32_t *)calloc(e->bufLen, sizeof(int32_t));
    e->mixV = (int32_t *)calloc(e->bufLen, sizeof(int32_t));
    e->predU = (int32_t *)calloc(e->bufLen, sizeof(int32_t));
    e->predV = (int32_t *)calloc(e->bufLen, sizeof(int32_t));
    e->shiftUV = (uint16_t *)calloc((size_t)e->bufLen * 2, sizeof(uint16_t));
    e->workBytes = oalac_max_packet_bytes(frameSize, bitDepth, numChannels) + 64;
    e->work = (uint8_t *)calloc(e->workBytes, 1);
    e->hooks = k_own_hooks;
    if (!e->mixU || !e->mixV || !e->predU || !e->predV || !e->shiftUV || !e->work) {
        oalac_encoder_free(e);
        return NULL;
    }
    oalac_encoder_reset_state(e);
    return e;
}

void oalac_encoder_free(oalac_encoder *e)
{
    if (!e) return;
    free(e->mixU);
    free(e->mixV);
    free(e->predU);
    free(e->predV);
    free(e->shiftUV);
    free(e->work);
    free(e);
}

void oalac_encoder_set_hooks(oalac_encoder *e, const oalac_hooks *h) { e->hooks = h ? *h : k_own_hooks; }

/* SetFastMode, ALACEncoder.h:44: stereo elements go through EncodeStereoFast (ALACEncoder.cu:998-1001) */
void oalac_encoder_set_fast_mode(oalac_encoder *e, int fast) { e->fastMode = fast ? 1 : 0; }

/* :1524-1531 */
void oalac_encoder_reset_state(oalac_encoder *e)
{
    for (int ch = 0; ch < OALAC_MAX_CHANNELS; ch++)
        for (int s = 0; s < OALAC_MAX_SEARCHES; s++) {
            oalac_init_coefs(e->coefsU[ch][s], OALAC_DENSHIFT, OALAC_MAX_COEFS);
            oalac_init_coefs(e->coefsV[ch][s], OALAC_DENSHIFT, OALAC_MAX_COEFS);
        }
}

void oalac_encoder_get_state(const oalac_encoder *e, int16_t *s)
{
    memcpy(s + 0, e->coefsU[0][3], 32);
    memcpy(s + 16, e->coefsU[0][7], 32);
    memcpy(s + 32, e->coefsV[0][3], 32);
    memcpy(s + 48, e->coefsV[0][7], 32);
}

void oalac_encoder_set_state(oalac_encoder *e, const int16_t *s)
{
    memcpy(e->coefsU[0][3], s + 0, 32);
    memcpy(e->coefsU[0][7], s + 16, 32);
    memcpy(e->coefsV[0][3], s + 32, 32);
    memcpy(e->coefsV[0][7], s + 48, 32);
}

void oalac_encoder_last_info(const oalac_encoder *e, uint32_t *info6) { memcpy(info6, e->info, sizeof(e->info)); }

/* GetConfig/GetMagicCookie, ALACEncoder.cu:1082-1140 (<= 2 channels: 24 bytes, big-endian fields).
 * maxFrameBytes/avgBitRate carry the encoder's running stats (0 when fetched before encoding,
 * as alacconvert does, convert-utility/main.cu:424-426). */
uint32_t oalac_magic_cookie(const oalac_encoder *e, uint8_t *c)
{
    uint32_t f = e->frameSize, mfb = e->maxFrameBytes, sr = e->sampleRate;
    c[0] = (uint8_t)(f >> 24); c[1] = (uint8_t)(f >> 16); c[2] = (uint8_t)(f >> 8); c[3] = (uint8_t)f;
    c[4] = 0;                   /* compatibleVersion */
    c[5] = (uint8_t)e->bitDepth;
    c[6] = OALAC_PB0;
    c[7] = OALAC_MB0;
    c[8] = OALAC_KB0;
    c[9] = (uint8_t)e->numChannels;
    c[10] = 0; c[11] = OALAC_MAX_RUN;
    c[12] = (uint8_t)(mfb >> 24); c[13] = (uint8_t)(mfb >> 16); c[14] = (uint8_t)(mfb >> 8); c[15] = (uint8_t)mfb;
    c[16] = c[17] = c[18] = c[19] = 0; /* avgBitRate: Finish() never computes it (:1064-1073) */
    c[20] = (uint8_t)(sr >> 24); c[21] = (uint8_t)(sr >> 16); c[22] = (uint8_t)(sr >> 8); c[23] = (uint8_t)sr;
    return 24;
}

/* GetMagicCookie for any channel count, ALACEncoder.cu:1109-1140: above 2 channels the 24-byte config is followed
 * by a 12-byte 'chan' atom header (size byte 24 in [3]) and an ALACAudioChannelLayout {tag, bitmap 0, descriptions 0}.
 * The fork stores the layout tag WITHOUT the big-endian swap (:1120 has no Swap32NtoB), i.e. in host order = little
 * endian here; restated as the fork does it.  buf must hold 48 bytes. */
uint32_t oalac_magic_cookie_full(const oalac_encoder *e, uint8_t *c)
{
    static const uint32_t tags[OALAC_MAX_CHANNELS] = {(100u << 16) | 1, (101u << 16) | 2, (113u << 16) | 3, (116u << 16) | 4,
                                                      (120u << 16) | 5, (124u << 16) | 6, (142u << 16) | 7, (127u << 16) | 8};
    oalac_magic_cookie(e, c);
    if (e->numChannels <= 2) return 24;
    memset(c + 24, 0, 24);
    c[27] = 24;
    memcpy(c + 28, "chan", 4);
    uint32_t t = tags[e->numChannels - 1];
    c[36] = (uint8_t)t; c[37] = (uint8_t)(t >> 8); c[38] = (uint8_t)(t >> 16); c[39] = (uint8_t)(t >> 24);
    return 48;
}

static inline uint32_t bytes_shifted_for(uint32_t bitDepth)
{
    return bitDepth == 32 ? 2 : bitDepth >= 24 ? 1 : 0; /* :327-332 */
}

/* EncodeStereoEscape, ALACEncoder.cu:749-806 */
static void encode_stereo_escape(oalac_encoder *e, uint8_t *out, uint64_t *pos, const uint8_t *pcm,
                                 uint32_t stride, uint32_t numSamples)
{
    uint32_t partial = (numSamples == e->frameSize) ? 0 : 1;
    uint32_t bps = bytes_per_sample(e->bitDepth);
    oalac_put_bits(out, pos, 0, 12);
    oalac_put_bits(out, pos, (partial << 3) | 1, 4);
    if (partial) oalac_put_bits(out, pos, numSamples, 32);
    for (uint32_t i = 0; i < numSamples * 2; i++) {
        /* 16/32: raw words; 20/24: de-interleave via mixNN(mixres 0, no shift) then bitDepth bits */
        int32_t x = load_sample(pcm + ((size_t)(i >> 1) * stride + (i & 1)) * bps, e->bitDepth);
        oalac_put_bits(out, pos, (uint32_t)x, e->bitDepth);
    }
}

/* EncodeStereo, ALACEncoder.cu:290-558 */
static int32_t encode_stereo(oalac_encoder *e, uint8_t *out, uint64_t *pos, const uint8_t *pcm,
                             uint32_t stride, uint32_t ch, uint32_t numSamples)
{
    uint64_t startPos = *pos;
    uint32_t bytesShifted = bytes_shifted_for(e->bitDepth);
    uint32_t chanBits = e->bitDepth - bytesShifted * 8 + 1;
    uint32_t partial = (numSamples == e->frameSize) ? 0 : 1;
    const int32_t mixBits = 2, maxRes = 4;
    const uint32_t pbFactor = 4, mode = 0;
    const uint32_t pb = (pbFactor * OALAC_PB0) / 4;
    uint32_t bits1, bits2;
    int32_t status;

    /* :353-379 mixRes search on the first numSamples/8 samples, all five passes walking row 7.
     * The fork reads those samples from the batched pre-mix (gpu_mixNN, :1144-1310): same math. */
    uint32_t minBits1 = 1u << 31, minBits2 = 1u << 31;
    int32_t bestRes = 0;
    uint32_t dilate = 8;
    for (int32_t mixRes = 0; mixRes <= maxRes; mixRes++) {
        /* the search-form kernels apply the shift but do not keep the low bytes (:1216-1310);
         * shiftUV written here is overwritten by the full mix below before anything reads it */
        mix_strided(pcm, stride, e->bitDepth, e->mixU, e->mixV, (int32_t)(numSamples / dilate), mixBits, mixRes,
                    e->shiftUV, (int32_t)bytesShifted);
        e->hooks.pc_block(e->mixU, e->predU, (int32_t)(numSamples / dilate), e->coefsU[ch][7], 8, chanBits, OALAC_DENSHIFT);
        e->hooks.pc_block(e->mixV, e->predV, (int32_t)(numSamples / dilate), e->coefsV[ch][7], 8, chanBits, OALAC_DENSHIFT);
        uint64_t wpos = 0;
        status = e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predU, e->work, &wpos, (int32_t)(numSamples / dilate), (int32_t)chanBits, &bits1);
        if (status) return status;
        status = e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predV, e->work, &wpos, (int32_t)(numSamples / dilate), (int32_t)chanBits, &bits2);
        if (status) return status;
        if (bits1 + bits2 < minBits1) {
            minBits1 = bits1 + bits2;
            bestRes = mixRes;
        }
    }
    int32_t mixRes = bestRes;

    /* :385-415 full mix with the chosen mixRes (+ shift-off bytes) */
    mix_strided(pcm, stride, e->bitDepth, e->mixU, e->mixV, (int32_t)numSamples, mixBits, mixRes, e->shiftUV,
                (int32_t)bytesShifted);

    /* :418-452 numUV search: 8 converge passes over numSamples/32, then dyn_comp over
     * numSamples/8 — entries [numSamples/32, numSamples/8) of the predictor buffers still hold
     * the mixRes = 4 search pass (the quirk of SURVEY §3.2; reproduced by buffer persistence) */
    uint32_t numU = 4, numV = 4;
    minBits1 = minBits2 = 1u << 31;
    for (uint32_t numUV = 4; numUV <= 8; numUV += 4) {
        dilate = 32;
        for (int converge = 0; converge < 8; converge++) {
            e->hooks.pc_block(e->mixU, e->predU, (int32_t)(numSamples / dilate), e->coefsU[ch][numUV - 1], (int32_t)numUV, chanBits, OALAC_DENSHIFT);
            e->hooks.pc_block(e->mixV, e->predV, (int32_t)(numSamples / dilate), e->coefsV[ch][numUV - 1], (int32_t)numUV, chanBits, OALAC_DENSHIFT);
        }
        dilate = 8;
        uint64_t wpos = 0;
        e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predU, e->work, &wpos, (int32_t)(numSamples / dilate), (int32_t)chanBits, &bits1);
        if (bits1 * dilate + 16 * numUV < minBits1) {
            minBits1 = bits1 * dilate + 16 * numUV;
            numU = numUV;
        }
        e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predV, e->work, &wpos, (int32_t)(numSamples / dilate), (int32_t)chanBits, &bits2);
        if (bits2 * dilate + 16 * numUV < minBits2) {
            minBits2 = bits2 * dilate + 16 * numUV;
            numV = numUV;
        }
    }

    /* :455-461 escape estimate */
    uint32_t minBits = minBits1 + minBits2 + 8 * 8 + (partial ? 32 : 0);
    if (bytesShifted != 0) minBits += numSamples * (bytesShifted * 8) * 2;
    uint32_t escapeBits = numSamples * e->bitDepth * 2 + (partial ? 32 : 0) + 2 * 8;
    int doEscape = (minBits >= escapeBits);

    e->info[0] = 0; e->info[1] = (uint32_t)mixRes; e->info[2] = numU; e->info[3] = numV; e->info[4] = e->info[5] = 0;

    if (!doEscape) {
        /* :466-485 header + coefficients as they stand BEFORE the final pass */
        oalac_put_bits(out, pos, 0, 12);
        oalac_put_bits(out, pos, (partial << 3) | (bytesShifted << 1), 4);
        if (partial) oalac_put_bits(out, pos, numSamples, 32);
        oalac_put_bits(out, pos, (uint32_t)mixBits, 8);
        oalac_put_bits(out, pos, (uint32_t)mixRes, 8);
        oalac_put_bits(out, pos, (mode << 4) | OALAC_DENSHIFT, 8);
        oalac_put_bits(out, pos, (pbFactor << 5) | numU, 8);
        for (uint32_t i = 0; i < numU; i++) oalac_put_bits(out, pos, (uint32_t)(int32_t)e->coefsU[ch][numU - 1][i], 16);
        oalac_put_bits(out, pos, (mode << 4) | OALAC_DENSHIFT, 8);
        oalac_put_bits(out, pos, (pbFactor << 5) | numV, 8);
        for (uint32_t i = 0; i < numV; i++) oalac_put_bits(out, pos, (uint32_t)(int32_t)e->coefsV[ch][numV - 1][i], 16);

        /* :488-500 interleaved shift-off bytes */
        if (bytesShifted != 0) {
            uint32_t bitShift = bytesShifted * 8;
            for (uint32_t i = 0; i < numSamples * 2; i += 2) {
                uint32_t val = ((uint32_t)e->shiftUV[i] << bitShift) | (uint32_t)e->shiftUV[i + 1];
                oalac_put_bits(out, pos, val, bitShift * 2);
            }
        }

        /* :505-532 final predictor + entropy coder, U then V into the same bit cursor */
        e->hooks.pc_block(e->mixU, e->predU, (int32_t)numSamples, e->coefsU[ch][numU - 1], (int32_t)numU, chanBits, OALAC_DENSHIFT);
        status = e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predU, out, pos, (int32_t)numSamples, (int32_t)chanBits, &bits1);
        if (status) return status;
        e->hooks.pc_block(e->mixV, e->predV, (int32_t)numSamples, e->coefsV[ch][numV - 1], (int32_t)numV, chanBits, OALAC_DENSHIFT);
        status = e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predV, out, pos, (int32_t)numSamples, (int32_t)chanBits, &bits2);
        if (status) return status;
        e->info[4] = bits1; e->info[5] = bits2;

        /* :537-543 compressed packet not smaller than an escape packet: rewind */
        minBits = (uint32_t)(*pos - startPos);
        if (minBits >= escapeBits) {
            *pos = startPos;
            doEscape = 1;
        }
    }
    if (doEscape) {
        e->info[0] = 1;
        encode_stereo_escape(e, out, pos, pcm, stride, numSamples);
    }
    return OALAC_noErr;
}

/* EncodeStereoFast, ALACEncoder.cu:564-745: no search — mixRes = kDefaultMixRes (0), numU = numV = kDefaultNumUV (8),
 * one pc_block pass over the whole packet on row 7 (whose coefficients, as they stand BEFORE the pass, go into the
 * header), escape decided from the bits actually written.  (In the fork the call site hands it the HOST buffer while its
 * mixNN are device kernels, :1001 — the path cannot run there; what is restated is Apple's function as written.) */
static int32_t encode_stereo_fast(oalac_encoder *e, uint8_t *out, uint64_t *pos, const uint8_t *pcm,
                                  uint32_t stride, uint32_t ch, uint32_t numSamples)
{
    uint64_t startPos = *pos;
    uint32_t bytesShifted = bytes_shifted_for(e->bitDepth); /* :596-603 */
    uint32_t chanBits = e->bitDepth - bytesShifted * 8 + 1;
    uint32_t partial = (numSamples == e->frameSize) ? 0 : 1;
    const int32_t mixBits = 2, mixRes = 0;                  /* :613-614 */
    const uint32_t numU = 8, numV = 8, pbFactor = 4, mode = 0; /* :615-618 */
    const uint32_t pb = (pbFactor * OALAC_PB0) / 4;
    uint32_t bits1 = 0, bits2 = 0;
    int32_t status;

    /* :623-644 */
    mix_strided(pcm, stride, e->bitDepth, e->mixU, e->mixV, (int32_t)numSamples, mixBits, mixRes, e->shiftUV,
                (int32_t)bytesShifted);
    /* :649-671 header + coefficients */
    oalac_put_bits(out, pos, 0, 12);
    oalac_put_bits(out, pos, (partial << 3) | (bytesShifted << 1), 4);
    if (partial) oalac_put_bits(out, pos, numSamples, 32);
    oalac_put_bits(out, pos, (uint32_t)mixBits, 8);
    oalac_put_bits(out, pos, (uint32_t)mixRes, 8);
    oalac_put_bits(out, pos, (mode << 4) | OALAC_DENSHIFT, 8);
    oalac_put_bits(out, pos, (pbFactor << 5) | numU, 8);
    for (uint32_t i = 0; i < numU; i++) oalac_put_bits(out, pos, (uint32_t)(int32_t)e->coefsU[ch][numU - 1][i], 16);
    oalac_put_bits(out, pos, (mode << 4) | OALAC_DENSHIFT, 8);
    oalac_put_bits(out, pos, (pbFactor << 5) | numV, 8);
    for (uint32_t i = 0; i < numV; i++) oalac_put_bits(out, pos, (uint32_t)(int32_t)e->coefsV[ch][numV - 1][i], 16);
    /* :674-686 */
    if (bytesShifted != 0) {
        uint32_t bitShift = bytesShifted * 8;
        for (uint32_t i = 0; i < numSamples * 2; i += 2) {
            uint32_t val = ((uint32_t)e->shiftUV[i] << bitShift) | (uint32_t)e->shiftUV[i + 1];
            oalac_put_bits(out, pos, val, bitShift * 2);
        }
    }
    /* :690-702 */
    e->hooks.pc_block(e->mixU, e->predU, (int32_t)numSamples, e->coefsU[ch][numU - 1], (int32_t)numU, chanBits, OALAC_DENSHIFT);
    status = e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predU, out, pos, (int32_t)numSamples, (int32_t)chanBits, &bits1);
    if (status) return status;
    e->hooks.pc_block(e->mixV, e->predV, (int32_t)numSamples, e->coefsV[ch][numV - 1], (int32_t)numV, chanBits, OALAC_DENSHIFT);
    status = e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predV, out, pos, (int32_t)numSamples, (int32_t)chanBits, &bits2);
    if (status) return status;
    /* :705-716 */
    uint32_t minBits = (bits1 + numU * 16) + (bits2 + numV * 16) + 8 * 8 + (partial ? 32 : 0);
    if (bytesShifted != 0) minBits += numSamples * (bytesShifted * 8) * 2;
    uint32_t escapeBits = numSamples * e->bitDepth * 2 + (partial ? 32 : 0) + 2 * 8;
    int doEscape = (minBits >= escapeBits);
    /* :718-729 */
    if (!doEscape && (uint32_t)(*pos - startPos) >= escapeBits) doEscape = 1;
    e->info[0] = (uint32_t)doEscape; e->info[1] = 0; e->info[2] = numU; e->info[3] = numV; e->info[4] = bits1; e->info[5] = bits2;
    if (doEscape) { /* :731-742 */
        *pos = startPos;
        encode_stereo_escape(e, out, pos, pcm, stride, numSamples);
    }
    return OALAC_noErr;
}

/* mono input widening: gpu_copyNNToPredictor, ALACEncoder.cu:1312-1382 (the fork's indexing of
 * those kernels is broken; the intended per-sample math is what is restated) */
static void copy_to_predictor(const uint8_t *pcm, uint32_t stride, uint32_t bitDepth, int32_t *out, uint16_t *shiftBuf,
                              uint32_t numSamples, uint32_t bytesShifted)
{
    uint32_t bps = bytes_per_sample(bitDepth);
    uint32_t shift = bytesShifted * 8;
    uint32_t mask = (1u << shift) - 1;
    for (uint32_t z = 0; z < numSamples; z++) {
        int32_t val = load_sample(pcm + (size_t)z * stride * bps, bitDepth);
        if (bytesShifted) {
            shiftBuf[z] = (uint16_t)((uint32_t)val & mask);
            val >>= shift;
        }
        out[z] = val;
    }
}

/* EncodeMono, ALACEncoder.cu:812-963.  The fork lost the `if (doEscape)` block after :959; the
 * escape packet restated here is the one the reference's own decoder parses
 * (ALACDecoder.cu:697-727): 12b 0, 4b (partial<<3)|1, [32b N], N samples of bitDepth bits. */
static int32_t encode_mono(oalac_encoder *e, uint8_t *out, uint64_t *pos, const uint8_t *pcm,
                           uint32_t stride, uint32_t ch, uint32_t numSamples)
{
    uint64_t startPos = *pos;
    uint32_t bytesShifted = bytes_shifted_for(e->bitDepth);
    uint32_t shift = bytesShifted * 8;
    uint32_t chanBits = e->bitDepth - bytesShifted * 8;
    uint32_t partial = (numSamples == e->frameSize) ? 0 : 1;
    const uint32_t pbFactor = 4;
    const uint32_t pb = (pbFactor * OALAC_PB0) / 4;
    uint32_t bits1;
    int32_t status;

    copy_to_predictor(pcm, stride, e->bitDepth, e->mixU, e->shiftUV, numSamples, bytesShifted);

    /* :874-905 */
    uint32_t minBits = 1u << 31, bestU = 4;
    for (uint32_t numU = 4; numU <= 8; numU += 4) {
        uint32_t dilate = 32;
        for (int converge = 0; converge < 7; converge++)
            e->hooks.pc_block(e->mixU, e->predU, (int32_t)(numSamples / dilate), e->coefsU[ch][numU - 1], (int32_t)numU, chanBits, OALAC_DENSHIFT);
        dilate = 8;
        e->hooks.pc_block(e->mixU, e->predU, (int32_t)(numSamples / dilate), e->coefsU[ch][numU - 1], (int32_t)numU, chanBits, OALAC_DENSHIFT);
        uint64_t wpos = 0;
        status = e->hooks.dyn_comp(OALAC_MB0, pb, OALAC_KB0, e->predU, e->work, &wpos, (int32_t)(numSamples / dilate), (int32_t)chanBits, &bits1);
        if (status) return status;
        uint32_t numBits = dilate * bits1 + 16 * numU;
        if (numBits < minBits) {
            bestU = numU;
            minBits = numBits;
        }
    }

    /* :907-915 */
    minBits += 4 * 8 + (partial ? 32 : 0);
    if (bytesShifted != 0) minBits += numSamples * (bytesShifted * 8);
    uint32_t escapeBits = numSamples * e->bitDepth + (partial ? 32 : 0) + 2 * 8;
    int doEscape = (minBits >= escapeBits);

    e->info[0] = 0; e->info[1] = 0; e->info[2] = bestU; e->info[3] = 0; e->info[4] = e->info[5] = 0;

    if (!doEscape) {
        /* :919-931 */
        uint32_t numU = bestU;
        oalac_put_bits(out, pos, 0, 12);
        oalac_put_bits(out, pos, (partial << 3) | (bytesShifted << 1), 4);
        if (partial) oalac_put_bits(out, pos, numSamples, 32);
        oalac_put_bits(out, pos, 0, 16);
        oalac_put_bits(out, pos, (0u << 4) | OALAC_DENSHIFT, 8);
        oalac_put_bits(out, pos, (pbFactor << 5) | numU, 8);
        for (uint32_t i = 0; i < numU; i++) oalac_put_bits(out, pos, (uint32_t)(int32_t)e->coefsU[ch][numU - 1][i], 16);
        /* :934-938 */
        if (bytesShifted != 0)
            for (uint32_t i = 0; i < numSamples; i++) oalac_put_bits(out, pos, e->shiftUV[i], shift);
        /* :941-945 (set_standard_ag_params == MB0, PB0, KB0) */
        e->hooks.pc_block(e->mixU, e->predU, (int32_t)numSamples, e->coefsU[ch][numU - 1], (int32_t)numU, chanBits, OALAC_DENSHIFT);
        status = e->hooks.dyn_comp(OALAC_MB0, OALAC_PB0, OALAC_KB0, e->predU, out, pos, (int32_t)numSamples, (int32_t)chanBits, &bits1);
        if (status) return status;
        e->info[4] = bits1;
        /* :952-958 */
        minBits = (uint32_t)(*pos - startPos);
        if (minBits >= escapeBits) {
            *pos = startPos;
            doEscape = 1;
        }
    }
    if (doEscape) {
        e->info[0] = 1;
        uint32_t bps = bytes_per_sample(e->bitDepth);
        oalac_put_bits(out, pos, 0, 12);
        oalac_put_bits(out, pos, (partial << 3) | 1, 4);
        if (partial) oalac_put_bits(out, pos, numSamples, 32);
        for (uint32_t i = 0; i < numSamples; i++)
            oalac_put_bits(out, pos, (uint32_t)load_sample(pcm + (size_t)i * stride * bps, e->bitDepth), e->bitDepth);
    }
    return OALAC_noErr;
}

/* sChannelMaps, ALACEncoder.cu:97-107 */
static const uint32_t k_channel_maps[OALAC_MAX_CHANNELS] = {
    0,
    1,
    (1u << 3) | 0,
    (0u << 9) | (1u << 3) | 0,
    (1u << 9) | (1u << 3) | 0,
    (0u << 15) | (1u << 9) | (1u << 3) | 0,
    (0u << 18) | (0u << 15) | (1u << 9) | (1u << 3) | 0,
    (0u << 21) | (1u << 15) | (1u << 9) | (1u << 3) | 0,
};

uint32_t oalac_channel_map(uint32_t numChannels)
{
    return (numChannels >= 1 && numChannels <= OALAC_MAX_CHANNELS) ? k_channel_maps[numChannels - 1] : 0;
}

/* Encode, ALACEncoder.cu:973-1057 */
int32_t oalac_encode_packet(oalac_encoder *e, const uint8_t *pcm, uint32_t numSamples, uint8_t *out,
                            uint32_t *outBytes)
{
    uint64_t pos = 0;
    int32_t status;
    if (numSamples > e->frameSize) return OALAC_ParamError;
    if (e->numChannels == 2) {
        oalac_put_bits(out, &pos, 1 /* ID_CPE */, 3);
        oalac_put_bits(out, &pos, 0, 4);
        status = e->fastMode ? encode_stereo_fast(e, out, &pos, pcm, 2, 0, numSamples)
                             : encode_stereo(e, out, &pos, pcm, 2, 0, numSamples); /* :998-1001 */
    } else if (e->numChannels == 1) {
        oalac_put_bits(out, &pos, 0 /* ID_SCE */, 3);
        oalac_put_bits(out, &pos, 0, 4);
        status = encode_mono(e, out, &pos, pcm, 1, 0, numSamples);
    } else {
        /* > 2 channels: the element loop of Apple's encoder that the fork dropped, driven by the
         * fork's own table sChannelMaps (ALACEncoder.cu:97-107: 3 bits per channel index, ID_SCE = 0 / ID_CPE = 1;
         * the LFE positions of the 5.1 .. 7.1 layouts carry ID_SCE in that table).  Every element type counts its
         * own 4-bit instance tag; element k uses the coefficient rows of its first channel and reads its samples
         * at that channel of the numChannels-interleaved frame.  Checked against the decoder's element loop
         * (ALACDecoder.cu:600-990), which is the reference code that parses it. */
        uint32_t bps = bytes_per_sample(e->bitDepth);
        uint32_t monoTag = 0, stereoTag = 0;
        status = OALAC_noErr;
        for (uint32_t ci = 0; ci < e->numChannels && status == OALAC_noErr;) {
            uint32_t tag = (k_channel_maps[e->numChannels - 1] >> (ci * 3)) & 7u;
            oalac_put_bits(out, &pos, tag, 3);
            if (tag == 1) {
                oalac_put_bits(out, &pos, stereoTag++, 4);
                /* mFastMode is consulted in the mChannelsPerFrame == 2 branch only (ALACEncoder.cu:998-1001): the element
                 * loop of a > 2-channel stream always searches */
                status = encode_stereo(e, out, &pos, pcm + (size_t)ci * bps, e->numChannels, ci, numSamples);
                ci += 2;
            } else {
                oalac_put_bits(out, &pos, monoTag++, 4);
                status = encode_mono(e, out, &pos, pcm + (size_t)ci * bps, e->numChannels, ci, numSamples);
                ci += 1;
            }
        }
    }
    if (status) return status;
    oalac_put_bits(out, &pos, 7 /* ID_END */, 3);
    if (pos & 7) oalac_put_bits(out, &pos, 0, 8 - (uint32_t)(pos & 7)); /* BitBufferByteAlign(true) */
    uint32_t n = (uint32_t)(pos / 8);
    *outBytes = n;
    e->totalBytes += n;
    if (n > e->maxFrameBytes) e->maxFrameBytes = n;
    return OALAC_noErr;
}

int64_t oalac_encode_stream(oalac_encoder *e, const uint8_t *pcm, uint64_t totalSamples,
                            uint32_t segmentPackets, uint8_t *out, uint64_t outCap,
                            uint32_t *packetBytes)
{
    uint32_t bytesPerFrame = e->numChannels * bytes_per_sample(e->bitDepth);
    uint32_t maxPkt = oalac_max_packet_bytes(e->frameSize, e->bitDepth, e->numChannels);
    uint8_t *tmp = (uint8_t *)calloc(maxPkt + 64, 1);
    if (!tmp) return OALAC_MemFullError;
    uint64_t done = 0, outPos = 0;
    uint32_t p = 0;
    while (done < totalSamples) {
        uint32_t n = (uint32_t)((totalSamples - done) < e->frameSize ? (totalSamples - done) : e->frameSize);
        if (segmentPackets && (p % segmentPackets) == 0) oalac_encoder_reset_state(e);
        uint32_t nb = 0;
        int32_t st = oalac_encode_packet(e, pcm + done * bytesPerFrame, n, tmp, &nb);
        if (st) { free(tmp); return st; }
        if (outPos + nb > outCap) { free(tmp); return OALAC_MemFullError; }
        memcpy(out + outPos, tmp, nb);
        packetBytes[p++] = nb;
        outPos += nb;
        done += n;
    }
    free(tmp);
    return (int64_t)outPos;
}

/* ------------------------------------------------------------------------------------------
 * decoder driver — codec/ALACDecoder.cu
 * ---------------------------------------------------------------------------------------- */

struct oalac_decoder {
    uint32_t frameLength, bitDepth, pb, mb, kb, numChannels, maxRun, sampleRate;
    int32_t *mixU, *mixV, *pred;
    uint16_t *shiftBuf;
    oalac_hooks hooks;
};

static inline uint32_t rd_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* Init, ALACDecoder.cu:109-190 */
oalac_decoder *oalac_decoder_new(const uint8_t *cookie, uint32_t size, int32_t *status)
{
    int32_t st = OALAC_noErr;
    oalac_decoder *d = NULL;
    if (size >= 12 && cookie[4] == 'f' && cookie[5] == 'r' && cookie[6] == 'm' && cookie[7] == 'a') { cookie += 12; size -= 12; }
    if (size >= 12 && cookie[4] == 'a' && cookie[5] == 'l' && cookie[6] == 'a' && cookie[7] == 'c') { cookie += 12; size -= 12; }
    if (size < 24) { st = OALAC_ParamError; goto done; }
    if (cookie[4] > 0) { st = OALAC_ParamError; goto done; } /* compatibleVersion <= kALACVersion */
    d = (oalac_decoder *)calloc(1, sizeof(*d));
    if (!d) { st = OALAC_MemFullError; goto done; }
    d->frameLength = rd_be32(cookie);
    d->bitDepth = cookie[5];
    d->pb = cookie[6];
    d->mb = cookie[7];
    d->kb = cookie[8];
    d->numChannels = cookie[9];
    d->maxRun = ((uint32_t)cookie[10] << 8) | cookie[11];
    d->sampleRate = rd_be32(cookie + 20);
    d->mixU = (int32_t *)calloc(d->frameLength + 64, sizeof(int32_t));
    d->mixV = (int32_t *)calloc(d->frameLength + 64, sizeof(int32_t));
    d->pred = (int32_t *)calloc(d->frameLength + 64, sizeof(int32_t));
    d->shiftBuf = (uint16_t *)calloc((size_t)(d->frameLength + 64) * 2, sizeof(uint16_t));
    d->hooks = k_own_hooks;
    if (!d->mixU || !d->mixV || !d->pred || !d->shiftBuf) { oalac_decoder_free(d); d = NULL; st = OALAC_MemFullError; }
done:
    if (status) *status = st;
    return d;
}

void oalac_decoder_free(oalac_decoder *d)
{
    if (!d) return;
    free(d->mixU);
    free(d->mixV);
    free(d->pred);
    free(d->shiftBuf);
    free(d);
}

void oalac_decoder_set_hooks(oalac_decoder *d, const oalac_hooks *h) { d->hooks = h ? *h : k_own_hooks; }

/* read one channel's header + samples of a compressed element (shared by SCE :656-696 and CPE :791-855) */
static int32_t decode_channel(oalac_decoder *d, uint8_t *pkt, uint32_t pktBytes, uint64_t *pos,
                              int32_t *dst, uint32_t numSamples, uint32_t chanBits, uint32_t mode,
                              uint32_t denShift, uint32_t pbFactor, int16_t *coefs, uint32_t num)
{
    uint32_t bits;
    int32_t st = d->hooks.dyn_decomp(d->mb, (d->pb * pbFactor) / 4, d->kb, pkt, pktBytes, pos, d->pred,
                                     (int32_t)numSamples, (int32_t)chanBits, &bits);
    if (st) return st;
    if (mode != 0) d->hooks.unpc_block(d->pred, d->pred, (int32_t)numSamples, NULL, 31, chanBits, 0);
    d->hooks.unpc_block(d->pred, dst, (int32_t)numSamples, coefs, (int32_t)num, chanBits, denShift);
    return OALAC_noErr;
}

/* Decode, ALACDecoder.cu:571-1002 + fillWriteBuffer :497-563 (un-mix / pack of this packet) */
int32_t oalac_decode_packet(oalac_decoder *d, const uint8_t *packet, uint32_t pktBytes, uint8_t *pcmOut,
                            uint32_t *outNumSamples)
{
    uint8_t *pkt = (uint8_t *)packet;
    uint64_t pos = 0;
    uint32_t numSamples = d->frameLength;
    uint32_t channelIndex = 0;
    uint32_t numChannels = d->numChannels;
    int16_t coefsU[32], coefsV[32];
    uint32_t bps = bytes_per_sample(d->bitDepth);
    int32_t status = OALAC_noErr;
    *outNumSamples = 0;

    while (status == OALAC_noErr) {
        if (!((pos >> 3) < pktBytes)) return OALAC_ParamError; /* :615 */
        uint32_t tag = oalac_get_bits(pkt, pktBytes, &pos, 3);
        switch (tag) {
        case 0: /* ID_SCE */
        case 3: /* ID_LFE */ {
            (void)oalac_get_bits(pkt, pktBytes, &pos, 4);
            if (oalac_get_bits(pkt, pktBytes, &pos, 12) != 0) return OALAC_ParamError;
            uint32_t hb = oalac_get_bits(pkt, pktBytes, &pos, 4);
            uint32_t partial = hb >> 3, bytesShifted = (hb >> 1) & 3, escapeFlag = hb & 1;
            if (bytesShifted == 3) return OALAC_ParamError;
            uint32_t chanBits = d->bitDepth - bytesShifted * 8;
            if (partial) numSamples = oalac_get_bits(pkt, pktBytes, &pos, 32);
            if (numSamples > d->frameLength) return OALAC_ParamError;
            if (!escapeFlag) {
                (void)oalac_get_bits(pkt, pktBytes, &pos, 16); /* mixBits, mixRes */
                hb = oalac_get_bits(pkt, pktBytes, &pos, 8);
                uint32_t modeU = hb >> 4, denShiftU = hb & 0xf;
                hb = oalac_get_bits(pkt, pktBytes, &pos, 8);
                uint32_t pbFactorU = hb >> 5, numU = hb & 0x1f;
                for (uint32_t i = 0; i < numU; i++) coefsU[i] = (int16_t)oalac_get_bits(pkt, pktBytes, &pos, 16);
                uint64_t shiftPos = pos;
                if (bytesShifted) pos += (uint64_t)bytesShifted * 8 * numSamples;
                status = decode_channel(d, pkt, pktBytes, &pos, d->mixU, numSamples, chanBits, modeU, denShiftU, pbFactorU, coefsU, numU);
                if (status) return status;
                if (bytesShifted)
                    for (uint32_t i = 0; i < numSamples; i++) d->shiftBuf[i] = (uint16_t)oalac_get_bits(pkt, pktBytes, &shiftPos, bytesShifted * 8);
            } else {
                uint32_t sh = 32 - chanBits;
                for (uint32_t i = 0; i < numSamples; i++)
                    d->mixU[i] = (int32_t)(oalac_get_bits(pkt, pktBytes, &pos, chanBits) << sh) >> sh;
                bytesShifted = 0;
            }
            /* gpu_copyPredictorTo16/20/24/24Shift/32/32Shift, ALACDecoder.cu:385-495 */
            for (uint32_t i = 0; i < numSamples; i++) {
                int32_t val = d->mixU[i];
                /* only gpu_copyPredictorTo24Shift / To32Shift re-attach the shifted-off bytes (fillWriteBuffer :503-531);
                 * the 16- and 20-bit routines write the predictor value as it is */
                if (bytesShifted && d->bitDepth >= 24) val = (int32_t)(((uint32_t)val << (bytesShifted * 8)) | d->shiftBuf[i]);
                store_sample(pcmOut + ((size_t)i * numChannels + channelIndex) * bps, d->bitDepth, val);
            }
            channelIndex += 1;
            *outNumSamples = numSamples;
            break;
        }
        case 1: /* ID_CPE */ {
            if (channelIndex + 2 > numChannels) goto no_more;
            (void)oalac_get_bits(pkt, pktBytes, &pos, 4);
            if (oalac_get_bits(pkt, pktBytes, &pos, 12) != 0) return OALAC_ParamError;
            uint32_t hb = oalac_get_bits(pkt, pktBytes, &pos, 4);
            uint32_t partial = hb >> 3, bytesShifted = (hb >> 1) & 3, escapeFlag = hb & 1;
            if (bytesShifted == 3) return OALAC_ParamError;
            uint32_t chanBits = d->bitDepth - bytesShifted * 8 + 1;
            if (partial) numSamples = oalac_get_bits(pkt, pktBytes, &pos, 32);
            if (numSamples > d->frameLength) return OALAC_ParamError;
            int32_t mixBits = 0, mixRes = 0;
            if (!escapeFlag) {
                mixBits = (int32_t)oalac_get_bits(pkt, pktBytes, &pos, 8);
                mixRes = (int8_t)oalac_get_bits(pkt, pktBytes, &pos, 8);
                hb = oalac_get_bits(pkt, pktBytes, &pos, 8);
                uint32_t modeU = hb >> 4, denShiftU = hb & 0xf;
                hb = oalac_get_bits(pkt, pktBytes, &pos, 8);
                uint32_t pbFactorU = hb >> 5, numU = hb & 0x1f;
                for (uint32_t i = 0; i < numU; i++) coefsU[i] = (int16_t)oalac_get_bits(pkt, pktBytes, &pos, 16);
                hb = oalac_get_bits(pkt, pktBytes, &pos, 8);
                uint32_t modeV = hb >> 4, denShiftV = hb & 0xf;
                hb = oalac_get_bits(pkt, pktBytes, &pos, 8);
                uint32_t pbFactorV = hb >> 5, numV = hb & 0x1f;
                for (uint32_t i = 0; i < numV; i++) coefsV[i] = (int16_t)oalac_get_bits(pkt, pktBytes, &pos, 16);
                uint64_t shiftPos = pos;
                if (bytesShifted) pos += (uint64_t)bytesShifted * 8 * 2 * numSamples;
                status = decode_channel(d, pkt, pktBytes, &pos, d->mixU, numSamples, chanBits, modeU, denShiftU, pbFactorU, coefsU, numU);
                if (status) return status;
                status = decode_channel(d, pkt, pktBytes, &pos, d->mixV, numSamples, chanBits, modeV, denShiftV, pbFactorV, coefsV, numV);
                if (status) return status;
                if (bytesShifted)
                    for (uint32_t i = 0; i < numSamples * 2; i++) d->shiftBuf[i] = (uint16_t)oalac_get_bits(pkt, pktBytes, &shiftPos, bytesShifted * 8);
            } else {
                chanBits = d->bitDepth;
                uint32_t sh = 32 - chanBits;
                for (uint32_t i = 0; i < numSamples; i++) {
                    d->mixU[i] = (int32_t)(oalac_get_bits(pkt, pktBytes, &pos, chanBits) << sh) >> sh;
                    d->mixV[i] = (int32_t)(oalac_get_bits(pkt, pktBytes, &pos, chanBits) << sh) >> sh;
                }
                bytesShifted = 0;
            }
            /* unmixNN(u, v, out + channelIndex, stride numChannels, ...): the fork's gpu_unmixNN handles the
             * stereo file only; the strided form is the upstream call the element loop was written for */
            unmix_strided(d->mixU, d->mixV, pcmOut + (size_t)channelIndex * bps, numChannels, d->bitDepth,
                          (int32_t)numSamples, mixBits, mixRes, d->shiftBuf, (int32_t)bytesShifted);
            channelIndex += 2;
            *outNumSamples = numSamples;
            break;
        }
        case 2: /* ID_CCE */
        case 5: /* ID_PCE */
            return OALAC_ParamError;
        case 4: /* ID_DSE, DataStreamElement :1033-1059 */ {
            (void)oalac_get_bits(pkt, pktBytes, &pos, 4);
            uint32_t align = oalac_get_bits(pkt, pktBytes, &pos, 1);
            uint32_t count = oalac_get_bits(pkt, pktBytes, &pos, 8);
            if (count == 255) count += oalac_get_bits(pkt, pktBytes, &pos, 8);
            if (align && (pos & 7)) pos += 8 - (pos & 7);
            pos += (uint64_t)count * 8;
            if ((pos + 7) / 8 > pktBytes) return OALAC_ParamError;
            break;
        }
        case 6: /* ID_FIL, FillElement :1012-1027 */ {
            int32_t count = (int32_t)oalac_get_bits(pkt, pktBytes, &pos, 4);
            if (count == 15) count += (int32_t)oalac_get_bits(pkt, pktBytes, &pos, 8) - 1;
            pos += (uint64_t)count * 8;
            if ((pos + 7) / 8 > pktBytes) return OALAC_ParamError;
            break;
        }
        case 7: /* ID_END */
            return status;
        }
        if (channelIndex >= numChannels) break;
    }
no_more:
    /* channels the packet did not carry are zero (ALACDecoder.cu:971-998; the fill is commented out in the fork,
     * upstream performs it) */
    for (; channelIndex < numChannels; channelIndex++)
        for (uint32_t i = 0; i < numSamples; i++)
            store_sample(pcmOut + ((size_t)i * numChannels + channelIndex) * bps, d->bitDepth, 0);
    return status;
}

uint64_t oalac_fnv1a64(const uint8_t *p, uint64_t n, uint64_t seed)
{
    uint64_t h = seed ? seed : 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
