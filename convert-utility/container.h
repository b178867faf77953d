/*
 * container.h — WAV / CAF reading and writing for alacconvert, in-memory (a file is a byte vector).
 *
 * Reproduces byte for byte what the reference's convert utility writes and accepts what it accepts:
 * convert-utility/main.cu:196-385 (format sniffing, data-chunk search), :387-643 (EncodeALAC file layout),
 * :646-790 (DecodeALAC), :803-852 (WAVE header), and convert-utility/CAFFileALAC.cpp:25-456 (chunk writers,
 * BER integers, packet-table header, chunk scanning).  The reference walks FILE*s with fseek/ftell; here the
 * layout is computed once and emitted in order, which gives the same bytes without any seeking.
 */
#ifndef ALACCONVERT_CONTAINER_H
#define ALACCONVERT_CONTAINER_H

#include <stdint.h>
#include <string>
#include <vector>

namespace alacfile {

typedef std::vector<uint8_t> Bytes;

enum FileKind { kUnknownFile = 0, kWaveFile, kCafFile, kM4aFile };

/* what the sniffers learn about an input file (main.cu:196-385, CAFFileALAC.cpp:395-456) */
struct InputInfo {
    FileKind kind;
    bool isAlac;          /* 'alac' (decode) vs 'lpcm' (encode) */
    bool bigEndianPcm;    /* CAF lpcm without the little-endian flag: samples are swapped on the way in */
    double sampleRate;
    uint32_t channels;
    uint32_t bitsPerChannel;   /* lpcm: sample width; alac: 0 */
    uint32_t alacSourceFlag;   /* alac: 1..4 = 16/20/24/32-bit source (desc.mFormatFlags) */
    uint32_t framesPerPacket;
    uint64_t dataPos;          /* first payload byte (past the CAF edit count) */
    uint64_t dataSize;         /* payload bytes, clamped to what the file really holds */
};

/* returns an empty string on success, else the reference's diagnostic text */
std::string sniff_input(const Bytes &file, InputInfo &info);

/* ---- BER-style variable length integers of the CAF packet table (CAFFileALAC.cpp:189-260) ---- */
void append_ber(Bytes &out, uint32_t value);
/* reads one integer from p[0..avail); *used = bytes consumed (0 on malformed input) */
uint32_t read_ber(const uint8_t *p, size_t avail, size_t *used);

/* ---- ALAC in CAF (encode side) ---- */
struct AlacCafParams {
    double sampleRate;
    uint32_t channels;
    uint32_t bitDepth;          /* 16 / 20 / 24 / 32 */
    uint32_t framesPerPacket;   /* 4096 */
    uint64_t inputDataBytes;    /* PCM payload size the packet-table header is derived from (BuildBasePacketTable) */
};
/* caff + desc + kuki (+ chan) + pakt (+ free) + data, exactly as main.cu:387-643 leaves the file */
Bytes build_alac_caf(const AlacCafParams &p, const Bytes &cookie, const std::vector<uint32_t> &packetBytes,
                     const uint8_t *stream, uint64_t streamBytes);

/* ---- ALAC in CAF (decode side) ---- */
struct AlacCafContents {
    Bytes cookie;
    std::vector<uint32_t> packetBytes;   /* packets the reference's read loop would decode (main.cu:719-737) */
    uint64_t dataPos;
    std::vector<uint64_t> packetPos;     /* M4A only: file offset of every packet (chunks need not be contiguous); empty for CAF */
};
std::string parse_alac_caf(const Bytes &file, const InputInfo &info, AlacCafContents &out);

/* ---- PCM out (decode side) ---- */
Bytes build_wave(double sampleRate, uint32_t channels, uint32_t bitsPerChannel, const uint8_t *pcm, uint64_t pcmBytes);
Bytes build_pcm_caf(double sampleRate, uint32_t channels, uint32_t bitsPerChannel, const uint8_t *pcm, uint64_t pcmBytes);

/* ---- the magic cookie outside CAF (ALACMagicCookieDescription.txt:177-238) ----
 * Legacy cookie: 'frma' format atom (12) + ALAC specific info header (12) + the cookie + terminator atom (8); what
 * ALACDecoder::Init skips in front of the config (codec/ALACDecoder.cu:123-134). */
Bytes wrap_legacy_cookie(const Bytes &cookie);
/* the bare cookie (24 or 48 bytes) out of a bare or legacy-wrapped one; empty if it is neither */
Bytes unwrap_cookie(const Bytes &cookie);
/* MP4/M4A: the SoundDescriptionBox 'stsd' with one 'alac' AudioSampleEntry (36 bytes), the 12-byte full-box header of
 * the ALAC specific info and the cookie as vended (:186-216); all fields big endian, sample rate as the 16.16 value
 * ISO/IEC 14496-12 defines for the entry (the cookie carries the exact rate). */
Bytes build_alac_sample_description(const Bytes &cookie, uint32_t channels, uint32_t bitsPerChannel, uint32_t sampleRate);
/* inverse: cookie + the entry's fields; returns "" or a diagnostic */
std::string parse_alac_sample_description(const Bytes &stsd, Bytes &cookie, uint32_t &channels, uint32_t &bitsPerChannel,
                                          uint32_t &sampleRate16_16);

/* ---- ALAC in MP4 / M4A (no reference counterpart beyond the sample description of ALACMagicCookieDescription.txt:
 * 177-216; box layout per ISO/IEC 14496-12) ----
 * build: 'ftyp' (M4A ), 'moov' { 'mvhd', 'trak' { 'tkhd', 'mdia' { 'mdhd', 'hdlr' soun, 'minf' { 'smhd', 'dinf' { 'dref' url },
 * 'stbl' { 'stsd' = build_alac_sample_description, 'stts', 'stsc', 'stsz', 'stco' | 'co64' } } } } }, 'mdat': one track, one
 * chunk holding every packet back to back; timescale = sample rate, one sample = one packet of framesPerPacket frames (the last
 * one shorter: totalFrames decides).  parse: any chunk layout (stsc runs, stco or co64, fixed or per-sample stsz). */
struct AlacM4aParams {
    uint32_t sampleRate, channels, bitDepth, framesPerPacket;
    uint64_t totalFrames;       /* valid sample-frames of the stream (the duration) */
};
Bytes build_alac_m4a(const AlacM4aParams &p, const Bytes &cookie, const std::vector<uint32_t> &packetBytes,
                     const uint8_t *stream, uint64_t streamBytes);
/* fills info (isAlac, channels, rate, alacSourceFlag from the sample size, framesPerPacket from stts) and out (cookie,
 * packetBytes, packetPos); returns "" or a diagnostic */
std::string parse_alac_m4a(const Bytes &file, InputInfo &info, AlacCafContents &out);
bool has_m4a_extension(const std::string &path);

/* byte order of CAF big-endian lpcm -> packed little-endian (main.cu:482-507) */
void swap_samples_in_place(uint8_t *pcm, uint64_t bytes, uint32_t bitsPerChannel);

/* file helpers */
bool read_file(const std::string &path, Bytes &out);
bool write_file(const std::string &path, const Bytes &data);
bool has_wav_extension(const std::string &path);   /* main.cu:792-808 */

}  // namespace alacfile

#endif
