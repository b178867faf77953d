/*
 * alac_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, single thread) of the reference's ALAC hot path.
 * It is the checker for the HIP path: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under alac_amd/ links, imports
 * or calls it; the product path never falls back to it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Parity pin: oracle/_ref (the reference's own C stage files
 * compiled by oracle/Makefile) + tests/golden fixtures generated from it.
 */
#ifndef ALAC_ORACLE_H
#define ALAC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: codec/ALACAudioTypes.h:54-60, codec/ALACBitUtilities.h:51-54 */
enum {
    OALAC_noErr = 0,
    OALAC_UnimplementedError = -4,
    OALAC_ParamError = -50,
    OALAC_MemFullError = -108
};

/* codec constants: aglib.h:36-47, dplib.h:41, ALACEncoder.cu:56-61 */
#define OALAC_QBSHIFT 9
#define OALAC_PB0 40
#define OALAC_MB0 10
#define OALAC_KB0 14
#define OALAC_MAX_RUN 255
#define OALAC_DENSHIFT 9
#define OALAC_MAX_COEFS 16
#define OALAC_MAX_SEARCHES 16
#define OALAC_MAX_CHANNELS 8

/* ---- stage functions (flat signatures; the bit cursor is a plain bit position) ---- */

/* codec/dp_enc.c:49-60 */
void oalac_init_coefs(int16_t *coefs, uint32_t denshift, int32_t numPairs);
/* codec/dp_enc.c:77-388 */
void oalac_pc_block(int32_t *in, int32_t *pc, int32_t num, int16_t *coefs, int32_t numactive,
                    uint32_t chanbits, uint32_t denshift);
/* codec/dp_dec.c:55-381 */
void oalac_unpc_block(int32_t *pc, int32_t *out, int32_t num, int16_t *coefs, int32_t numactive,
                      uint32_t chanbits, uint32_t denshift);
/* codec/ag_enc.c:249-367; buf must be writable for ceil(bits/8) bytes past *bitpos */
int32_t oalac_dyn_comp(uint32_t mb0, uint32_t pb, uint32_t kb, int32_t *pc, uint8_t *buf,
                       uint64_t *bitpos, int32_t numSamples, int32_t bitSize, uint32_t *outNumBits);
/* codec/ag_dec.c:272-362; bufbytes bounds every read (bytes past the end read as 0) */
int32_t oalac_dyn_decomp(uint32_t mb0, uint32_t pb, uint32_t kb, uint8_t *buf, uint64_t bufbytes,
                         uint64_t *bitpos, int32_t *pc, int32_t numSamples, int32_t maxSize,
                         uint32_t *outNumBits);

/* codec/matrix_enc.cu:72-425 (mix16/20/24/32); pcm = packed little-endian interleaved stereo */
void oalac_mix(const uint8_t *pcm, uint32_t bitDepth, int32_t *u, int32_t *v, int32_t numSamples,
               int32_t mixbits, int32_t mixres, uint16_t *shiftUV, int32_t bytesShifted);
/* codec/ALACDecoder.cu:193-383 (gpu_unmix16/20/24/32) */
void oalac_unmix(const int32_t *u, const int32_t *v, uint8_t *pcm, uint32_t bitDepth,
                 int32_t numSamples, int32_t mixbits, int32_t mixres, const uint16_t *shiftUV,
                 int32_t bytesShifted);

/* MSB-first bit writer / reader: codec/ALACBitUtilities.c:212-249, :42-65 */
void oalac_put_bits(uint8_t *buf, uint64_t *bitpos, uint32_t value, uint32_t numBits);
uint32_t oalac_get_bits(const uint8_t *buf, uint64_t bufbytes, uint64_t *bitpos, uint32_t numBits);

/* hooks so the drivers can be run over the reference's compiled stage objects (oracle/_ref) */
typedef struct oalac_hooks {
    void (*pc_block)(int32_t *, int32_t *, int32_t, int16_t *, int32_t, uint32_t, uint32_t);
    void (*unpc_block)(int32_t *, int32_t *, int32_t, int16_t *, int32_t, uint32_t, uint32_t);
    int32_t (*dyn_comp)(uint32_t, uint32_t, uint32_t, int32_t *, uint8_t *, uint64_t *, int32_t,
                        int32_t, uint32_t *);
    int32_t (*dyn_decomp)(uint32_t, uint32_t, uint32_t, uint8_t *, uint64_t, uint64_t *, int32_t *,
                          int32_t, int32_t, uint32_t *);
} oalac_hooks;

/* ---- encoder driver: codec/ALACEncoder.cu:290-558, :749-806, :812-963, :973-1057, :1457-1535 ---- */
typedef struct oalac_encoder oalac_encoder;

oalac_encoder *oalac_encoder_new(uint32_t frameSize, uint32_t bitDepth, uint32_t numChannels,
                                 uint32_t sampleRate);
void oalac_encoder_free(oalac_encoder *e);
void oalac_encoder_set_fast_mode(oalac_encoder *e, int fast);
void oalac_encoder_set_hooks(oalac_encoder *e, const oalac_hooks *h);
/* reset the persistent coefficient rows to init_coefs (start of an independent segment) */
void oalac_encoder_reset_state(oalac_encoder *e);
/* copy rows [search 3] and [search 7] of U and V (the only rows the search touches) in/out:
 * layout int16 [2 ch: U,V][2 rows: numUV 4, 8][16] */
void oalac_encoder_get_state(const oalac_encoder *e, int16_t *state64);
void oalac_encoder_set_state(oalac_encoder *e, const int16_t *state64);
/* one packet: pcm = numSamples sample-frames of packed LE interleaved PCM; returns status,
 * packet bytes in *outBytes.  out must hold oalac_max_packet_bytes(). */
int32_t oalac_encode_packet(oalac_encoder *e, const uint8_t *pcm, uint32_t numSamples, uint8_t *out,
                            uint32_t *outBytes);
uint32_t oalac_max_packet_bytes(uint32_t frameSize, uint32_t bitDepth, uint32_t numChannels);
/* debug/inspection of the last packet: [0]=escape [1]=mixRes [2]=numU [3]=numV [4]=bitsU [5]=bitsV */
void oalac_encoder_last_info(const oalac_encoder *e, uint32_t *info6);
/* 24-byte magic cookie: codec/ALACEncoder.cu:1082-1140 (<=2 channels) */
uint32_t oalac_magic_cookie(const oalac_encoder *e, uint8_t *cookie24);

/* 24 or (numChannels > 2) 48 bytes: config + 'chan' atom + channel layout, codec/ALACEncoder.cu:1109-1140 */
uint32_t oalac_magic_cookie_full(const oalac_encoder *e, uint8_t *cookie48);
/* sChannelMaps[numChannels - 1], codec/ALACEncoder.cu:97-107 (3 bits per channel index: 0 = ID_SCE, 1 = ID_CPE) */
uint32_t oalac_channel_map(uint32_t numChannels);

/* encode `numPackets` consecutive packets; if segmentPackets > 0 the state is reset every
 * segmentPackets packets (independent segments), 0 = one chained stream.  Sizes go to
 * packetBytes[numPackets]; packets are written back to back into out. Returns total bytes or <0. */
int64_t oalac_encode_stream(oalac_encoder *e, const uint8_t *pcm, uint64_t totalSamples,
                            uint32_t segmentPackets, uint8_t *out, uint64_t outCap,
                            uint32_t *packetBytes);

/* ---- decoder driver: codec/ALACDecoder.cu:109-190, :571-1002 ---- */
typedef struct oalac_decoder oalac_decoder;
oalac_decoder *oalac_decoder_new(const uint8_t *cookie, uint32_t cookieSize, int32_t *status);
void oalac_decoder_free(oalac_decoder *d);
void oalac_decoder_set_hooks(oalac_decoder *d, const oalac_hooks *h);
/* decode one packet to packed LE interleaved PCM (numChannels from the cookie) */
int32_t oalac_decode_packet(oalac_decoder *d, const uint8_t *packet, uint32_t packetBytes,
                            uint8_t *pcmOut, uint32_t *outNumSamples);

/* FNV-1a 64 over a byte range (fixture hashes) */
uint64_t oalac_fnv1a64(const uint8_t *p, uint64_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
