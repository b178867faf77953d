/*
 * container.cpp — see container.h.  Every layout decision cites the reference line it reproduces.
 */
#include "container.h"

#include <cctype>
#include <cstdio>
#include <cstring>
#include <utility>

namespace alacfile {

namespace {

const uint32_t kFramesPerPacketDefault = 4096;  // kALACDefaultFramesPerPacket, codec/ALACAudioTypes.h:74
const uint32_t kEscapeHeaderBytes = 8;          // kALACMaxEscapeHeaderBytes, codec/ALACAudioTypes.h:71
const uint32_t kPaktHeaderBytes = 24;           // kMinCAFFPacketTableHeaderSize, CAFFileALAC.h:32
const uint32_t kChunkHeaderBytes = 12;          // sizeof(port_CAFChunkHeader), CAFFileALAC.h:110-117

// layout tags by channel count (codec/ALACAudioTypes.h:100-126 == CAFFileALAC.h:36-60)
const uint32_t kLayoutTags[8] = {(100u << 16) | 1, (101u << 16) | 2, (113u << 16) | 3, (116u << 16) | 4,
                                 (120u << 16) | 5, (124u << 16) | 6, (142u << 16) | 7, (127u << 16) | 8};

inline bool tag_is(const uint8_t *p, const char *t) { return memcmp(p, t, 4) == 0; }
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t le32(const uint8_t *p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }

void put_tag(Bytes &o, const char *t) { o.insert(o.end(), t, t + 4); }
void put_be32(Bytes &o, uint32_t v)
{
    for (int s = 24; s >= 0; s -= 8) o.push_back((uint8_t)(v >> s));
}
void put_be64(Bytes &o, uint64_t v)
{
    for (int s = 56; s >= 0; s -= 8) o.push_back((uint8_t)(v >> s));
}
void put_le32(Bytes &o, uint32_t v)
{
    for (int s = 0; s < 32; s += 8) o.push_back((uint8_t)(v >> s));
}
void put_zeros(Bytes &o, size_t n) { o.insert(o.end(), n, 0); }
void put_be_f64(Bytes &o, double d)
{
    uint64_t u;
    memcpy(&u, &d, 8);
    put_be64(o, u);
}
double get_be_f64(const uint8_t *p)
{
    uint64_t u = ((uint64_t)be32(p) << 32) | be32(p + 4);
    double d;
    memcpy(&d, &u, 8);
    return d;
}

// 'caff', version 1, flags 0 (CAFFileALAC.cpp:60-64)
void put_caf_file_header(Bytes &o)
{
    put_tag(o, "caff");
    o.push_back(0);
    o.push_back(1);
    o.push_back(0);
    o.push_back(0);
}

// chunk header whose size the reference stores in the LAST byte only (desc, kuki, chan: CAFFileALAC.cpp:70, :107, :132)
void put_small_chunk_header(Bytes &o, const char *t, uint8_t size)
{
    put_tag(o, t);
    put_zeros(o, 7);
    o.push_back(size);
}

// CAFFileALAC.cpp:66-96
void put_desc(Bytes &o, double rate, uint32_t formatID, uint32_t flags, uint32_t bytesPerPacket, uint32_t framesPerPacket,
              uint32_t channels, uint32_t bits)
{
    put_small_chunk_header(o, "desc", 32);
    put_be_f64(o, rate);
    put_be32(o, formatID);
    put_be32(o, flags);
    put_be32(o, bytesPerPacket);
    put_be32(o, framesPerPacket);
    put_be32(o, channels);
    put_be32(o, bits);
}

// CAFFileALAC.cpp:128-139: tag, bitmap 0, no descriptions
void put_chan(Bytes &o, uint32_t tag)
{
    put_small_chunk_header(o, "chan", 12);
    put_be32(o, tag);
    put_zeros(o, 8);
}

// The reference sizes a sample as (mBitsPerChannel >> 3), main.cu:389 — 2 bytes for 20-bit material, which its own encoder
// reads as 3-byte containers (mix20, copy20ToPredictor): it never handled a 20-bit file.  Here a 20-bit sample takes the 3-byte
// container the codec reads (20 bits left-justified in 24, as WAVE and CAF store them); 16 / 24 / 32 bits are the reference's.
uint32_t bytes_per_sample(uint32_t bits) { return (bits + 7) >> 3; }

// Walk CAF chunks from offset 8 the way the reference's scanners do (12-byte header, size = low 32 bits of the
// big-endian int64, no bounds other than end of file): position of the first chunk of type `want`, or npos.
size_t find_caf_chunk(const Bytes &f, const char *want, uint32_t *sizeLow)
{
    size_t pos = 8;
    while (pos + kChunkHeaderBytes <= f.size()) {
        const uint32_t sz = be32(&f[pos + 8]);
        if (tag_is(&f[pos], want)) {
            if (sizeLow) *sizeLow = sz;
            return pos;
        }
        pos += kChunkHeaderBytes + (size_t)sz;
    }
    return (size_t)-1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// sniffing
// ---------------------------------------------------------------------------------------------
std::string sniff_input(const Bytes &f, InputInfo &info)
{
    memset(&info, 0, sizeof(info));
    const std::string cannot = "Cannot determine what format file is";
    if (f.size() < 12) return cannot;

    if (tag_is(&f[0], "caff")) {
        info.kind = kCafFile;
        // GetCAFFdescFormat (CAFFileALAC.cpp:395-456): chunks from offset 8, type(4) size(8)
        size_t pos = 8;
        bool found = false;
        while (pos + kChunkHeaderBytes <= f.size()) {
            if (tag_is(&f[pos], "desc")) {
                if (pos + kChunkHeaderBytes + 32 > f.size()) return cannot;
                const uint8_t *d = &f[pos + kChunkHeaderBytes];
                info.sampleRate = get_be_f64(d);
                const uint32_t formatID = be32(d + 8), flags = be32(d + 12);
                info.framesPerPacket = be32(d + 20);
                info.channels = be32(d + 24);
                info.bitsPerChannel = be32(d + 28);
                if (formatID == 0x616c6163u) {  // 'alac'
                    info.isAlac = true;
                    info.alacSourceFlag = flags;
                } else if (formatID == 0x6c70636du) {  // 'lpcm'
                    info.isAlac = false;
                    info.bigEndianPcm = (flags & 2u) == 0;  // CAF flag bit 1 = little endian (:428-436)
                } else {
                    return "data format is of an unsupported type";  // main.cu:145-149
                }
                found = true;
                break;
            }
            pos += kChunkHeaderBytes + (size_t)be32(&f[pos + 8]);
        }
        if (!found) return cannot;
        // FindCAFFDataStart (CAFFileALAC.cpp:363-393): payload starts past the 4-byte edit count
        uint32_t szLow = 0;
        const size_t dpos = find_caf_chunk(f, "data", &szLow);
        if (dpos == (size_t)-1 || dpos + kChunkHeaderBytes + 4 > f.size()) return "no data chunk";
        info.dataPos = dpos + kChunkHeaderBytes + 4;
        uint64_t sz = (uint64_t)szLow - 4;
        const uint64_t avail = f.size() - info.dataPos;
        if (szLow < 4 || sz > avail) sz = avail;  // unknown (-1) or overlong size: take what the file holds
        info.dataSize = sz;
        return "";
    }

    if (tag_is(&f[4], "ftyp")) {  // ISO base media file (MP4 / M4A): the whole parse is parse_alac_m4a's
        AlacCafContents c;
        return parse_alac_m4a(f, info, c);
    }

    if (tag_is(&f[0], "RIFF") && tag_is(&f[8], "WAVE")) {
        info.kind = kWaveFile;
        // GetInputFormat (main.cu:213-259): first 'fmt ' chunk, plain PCM only; other chunks skipped unpadded
        size_t pos = 12;
        bool found = false;
        while (pos + 8 <= f.size()) {
            if (tag_is(&f[pos], "fmt ")) {
                if (pos + 24 > f.size()) return cannot;
                const uint8_t *b = &f[pos + 4];  // the 20 bytes the reference reads: size(4) then the format fields
                if (b[4] != 1 || b[5] != 0) return cannot;  // only WAVE_FORMAT_PCM (:229-234)
                info.isAlac = false;
                info.channels = b[6];
                info.sampleRate = (double)le32(b + 8);
                info.bitsPerChannel = b[18];
                info.framesPerPacket = 1;
                found = true;
                break;
            }
            pos += 8 + (size_t)le32(&f[pos + 4]);
        }
        if (!found) return cannot;
        // FindDataStart (main.cu:336-363): chunks from 12 while inside the RIFF size
        const uint64_t riffSize = le32(&f[4]);
        pos = 12;
        bool have = false;
        while (pos < riffSize && pos + 8 <= f.size()) {
            const uint32_t sz = le32(&f[pos + 4]);
            if (tag_is(&f[pos], "data")) {
                info.dataPos = pos + 8;
                const uint64_t avail = f.size() - info.dataPos;
                info.dataSize = sz > avail ? avail : sz;
                have = true;
                break;
            }
            pos += 8 + (size_t)sz;
        }
        if (!have) return "no data chunk";
        return "";
    }
    return cannot;
}

// ---------------------------------------------------------------------------------------------
// BER integers
// ---------------------------------------------------------------------------------------------
void append_ber(Bytes &o, uint32_t v)
{
    // CAFFileALAC.cpp:189-236: 7 bits per byte, most significant group first, continuation bit on all but the last
    int groups = 1;
    while (groups < 5 && (v >> (7 * groups)) != 0) groups++;
    for (int g = groups - 1; g >= 0; g--) {
        uint8_t b = (uint8_t)((v >> (7 * g)) & 0x7f);
        if (g) b |= 0x80;
        o.push_back(b);
    }
}

uint32_t read_ber(const uint8_t *p, size_t avail, size_t *used)
{
    // CAFFileALAC.cpp:238-260 (gives up past 5 bytes)
    uint32_t v = 0;
    size_t n = 0;
    *used = 0;
    for (;;) {
        if (n >= avail || n >= 5) return 0;
        const uint8_t b = p[n++];
        v = (v << 7) | (b & 0x7f);
        if (!(b & 0x80)) break;
    }
    *used = n;
    return v;
}

// ---------------------------------------------------------------------------------------------
// ALAC in CAF, encode side
// ---------------------------------------------------------------------------------------------
Bytes build_alac_caf(const AlacCafParams &p, const Bytes &cookie, const std::vector<uint32_t> &packetBytes,
                     const uint8_t *stream, uint64_t streamBytes)
{
    Bytes o;
    o.reserve((size_t)streamBytes + 4096 + packetBytes.size() * 3);
    put_caf_file_header(o);
    // 'desc': alac, flags 1..4 by source depth, VBR (bytes per packet 0), no bits per channel (main.cu:268-301)
    const uint32_t depthFlag = p.bitDepth == 16 ? 1 : p.bitDepth == 20 ? 2 : p.bitDepth == 24 ? 3 : 4;
    put_desc(o, p.sampleRate, 0x616c6163u, depthFlag, 0, p.framesPerPacket, p.channels, 0);
    // 'kuki' (CAFFileALAC.cpp:105-112)
    put_small_chunk_header(o, "kuki", (uint8_t)cookie.size());
    o.insert(o.end(), cookie.begin(), cookie.end());
    if (p.channels > 2) put_chan(o, kLayoutTags[(p.channels - 1) & 7]);  // main.cu:423-427

    // BuildBasePacketTable (CAFFileALAC.cpp:262-287).  remainder = 4096 - (valid mod 4096) is never 0, so the
    // header always counts one packet more than valid/4096 — a phantom one when the input is an exact multiple.
    const uint32_t bps = bytes_per_sample(p.bitDepth);
    const int64_t validFrames = (int64_t)(p.inputDataBytes / ((uint64_t)bps * p.channels));
    int64_t numPackets = validFrames / kFramesPerPacketDefault;
    int32_t remainder = (int32_t)(validFrames - numPackets * kFramesPerPacketDefault);
    remainder = (int32_t)kFramesPerPacketDefault - remainder;
    if (remainder) numPackets += 1;
    const uint32_t maxPacket = bps * p.channels * kFramesPerPacketDefault + kEscapeHeaderBytes;
    const uint32_t entryBytes = maxPacket < 16384 ? 2 : 3;
    const uint64_t tableBytes = (uint64_t)entryBytes * (uint64_t)numPackets;

    Bytes entries;
    for (size_t i = 0; i < packetBytes.size(); i++) append_ber(entries, packetBytes[i]);
    // The reference reserves tableBytes, writes the entries into that space and, when more than a chunk header
    // is left over, turns the rest into a 'free' chunk and shrinks the 'pakt' size (main.cu:610-622).  Entries
    // never exceed the reservation for real packets (a packet is < 2^14 resp. 2^21 bytes).
    const uint64_t used = entries.size();
    const uint64_t left = tableBytes > used ? tableBytes - used : 0;
    const bool freeChunk = left > kChunkHeaderBytes;
    const uint64_t paktSize = (freeChunk ? used : (tableBytes > used ? tableBytes : used)) + kPaktHeaderBytes;

    put_tag(o, "pakt");
    if (freeChunk) {
        put_be64(o, paktSize);  // patched with WriteCAFFChunkSize: a full 8-byte field
    } else {
        put_zeros(o, 4);        // WriteCAFFpaktChunkHeader stores the low 32 bits only (CAFFileALAC.cpp:163-187)
        put_be32(o, (uint32_t)paktSize);
    }
    put_be64(o, (uint64_t)numPackets);
    put_be64(o, (uint64_t)validFrames);
    put_be32(o, 0);  // priming frames
    put_be32(o, (uint32_t)remainder);
    o.insert(o.end(), entries.begin(), entries.end());
    if (freeChunk) {
        // WriteCAFFfreeChunk (CAFFileALAC.cpp:141-161): header + zero fill, `left` bytes in total
        put_tag(o, "free");
        put_zeros(o, 4);
        put_be32(o, (uint32_t)(left - kChunkHeaderBytes));
        put_zeros(o, (size_t)(left - kChunkHeaderBytes));
    } else {
        put_zeros(o, (size_t)left);  // the unwritten part of the zero-filled reservation
    }
    // 'data': size = payload + the 4-byte edit count, edit count 1 (CAFFileALAC.cpp:98-103, main.cu:624-629)
    put_tag(o, "data");
    put_be64(o, streamBytes + 4);
    put_be32(o, 1);
    o.insert(o.end(), stream, stream + streamBytes);
    return o;
}

// ---------------------------------------------------------------------------------------------
// ALAC in CAF, decode side
// ---------------------------------------------------------------------------------------------
std::string parse_alac_caf(const Bytes &f, const InputInfo &info, AlacCafContents &out)
{
    out.cookie.clear();
    out.packetBytes.clear();
    out.dataPos = info.dataPos;
    // cookie: size = last byte of the chunk header (CAFFileALAC.cpp:289-361)
    const size_t kpos = find_caf_chunk(f, "kuki", nullptr);
    if (kpos == (size_t)-1) return "no kuki chunk";
    const size_t ksize = f[kpos + 11];
    if (kpos + kChunkHeaderBytes + ksize > f.size()) return "truncated kuki chunk";
    out.cookie.assign(f.begin() + kpos + kChunkHeaderBytes, f.begin() + kpos + kChunkHeaderBytes + ksize);
    // packet table entries start past the 24-byte table header (CAFFileALAC.cpp:25-58)
    const size_t ppos = find_caf_chunk(f, "pakt", nullptr);
    if (ppos == (size_t)-1) return "no pakt chunk";
    size_t tpos = ppos + kChunkHeaderBytes + kPaktHeaderBytes;
    // main.cu:706-737: sizes are read one BER integer at a time through a 5-byte window, with no regard for the
    // end of the table; the loop ends on a zero size or when the payload runs out.
    uint64_t dpos = info.dataPos;
    for (;;) {
        if (tpos >= f.size()) break;
        size_t used = 0;
        const size_t window = f.size() - tpos < 5 ? f.size() - tpos : 5;
        const uint32_t sz = read_ber(&f[tpos], window, &used);
        if (sz == 0 || used == 0) break;
        if (dpos + sz > f.size()) break;  // short fread
        out.packetBytes.push_back(sz);
        dpos += sz;
        tpos += used;
    }
    return "";
}

// ---------------------------------------------------------------------------------------------
// PCM out
// ---------------------------------------------------------------------------------------------
Bytes build_wave(double sampleRate, uint32_t channels, uint32_t bits, const uint8_t *pcm, uint64_t pcmBytes)
{
    // main.cu:803-852 + the two size patches at :766-772
    Bytes o;
    o.reserve((size_t)pcmBytes + 44);
    const uint32_t bytesPerFrame = channels * bytes_per_sample(bits);
    const uint32_t rate = (uint32_t)sampleRate;
    put_tag(o, "RIFF");
    put_le32(o, (uint32_t)(pcmBytes + 4 + 8 + 24));
    put_tag(o, "WAVE");
    put_tag(o, "fmt ");
    put_le32(o, 16);
    o.push_back(1);  // PCM
    o.push_back(0);
    o.push_back((uint8_t)channels);
    o.push_back(0);
    put_le32(o, rate);
    put_le32(o, rate * bytesPerFrame);
    o.push_back((uint8_t)bytesPerFrame);
    o.push_back(0);
    o.push_back((uint8_t)bits);
    o.push_back(0);
    put_tag(o, "data");
    put_le32(o, (uint32_t)pcmBytes);
    o.insert(o.end(), pcm, pcm + pcmBytes);
    return o;
}

// ---- the cookie outside CAF: ALACMagicCookieDescription.txt:177-238 ----

namespace {
const uint32_t kFrma = 0x66726d61u, kAlac = 0x616c6163u, kStsd = 0x73747364u;
void put_be16(Bytes &o, uint32_t v)
{
    o.push_back((uint8_t)(v >> 8));
    o.push_back((uint8_t)v);
}
}  // namespace

Bytes wrap_legacy_cookie(const Bytes &cookie)
{
    Bytes o;
    put_be32(o, 12);  // format atom
    put_be32(o, kFrma);
    put_be32(o, kAlac);
    put_be32(o, (uint32_t)(12 + cookie.size()));  // ALAC specific info: 36 = 12 + sizeof(ALACSpecificConfig)
    put_be32(o, kAlac);
    put_be32(o, 0);   // version / flags
    o.insert(o.end(), cookie.begin(), cookie.end());
    put_be32(o, 8);   // terminator atom
    put_be32(o, 0);
    return o;
}

Bytes unwrap_cookie(const Bytes &c)
{
    size_t pos = 0, end = c.size();
    bool wrapped = false;
    if (end - pos >= 12 && be32(&c[pos + 4]) == kFrma) pos += 12, wrapped = true;  // codec/ALACDecoder.cu:123-128
    if (end - pos >= 12 && be32(&c[pos + 4]) == kAlac) {                           // :129-134
        const uint32_t infoSize = be32(&c[pos]);
        pos += 12;
        if (infoSize >= 12 + 24 && pos + (infoSize - 12) <= end) end = pos + (infoSize - 12);
        wrapped = true;
    }
    if (!wrapped && end >= 8 && be32(&c[end - 8]) == 8 && be32(&c[end - 4]) == 0 && (end == 32 || end == 56)) end -= 8;
    const size_t n = end - pos;
    if (n < 24) return Bytes();
    // 24-byte config, or config + 'chan' atom + layout
    const size_t take = (n >= 48 && be32(&c[pos + 28]) == 0x6368616eu) ? 48 : 24;
    return Bytes(c.begin() + pos, c.begin() + pos + take);
}

Bytes build_alac_sample_description(const Bytes &cookie, uint32_t channels, uint32_t bits, uint32_t sampleRate)
{
    Bytes o;
    const uint32_t entry = 36 + 12 + (uint32_t)cookie.size();
    put_be32(o, 16 + entry);  // SoundDescriptionBox
    put_be32(o, kStsd);
    put_be32(o, 0);           // version / flags
    put_be32(o, 1);           // entry count
    put_be32(o, entry);       // AudioSampleEntry
    put_be32(o, kAlac);
    for (int i = 0; i < 6; i++) o.push_back(0);  // reserved
    put_be16(o, 1);           // data reference index
    put_be32(o, 0);           // reserved[2]
    put_be32(o, 0);
    put_be16(o, channels);
    put_be16(o, bits);
    put_be16(o, 0);           // predefined
    put_be16(o, 0);           // reserved
    put_be32(o, sampleRate << 16);  // 16.16 (ISO/IEC 14496-12 AudioSampleEntry); rates >= 65536 wrap, the cookie is exact
    put_be32(o, 12 + (uint32_t)cookie.size());  // ALAC specific info (full box)
    put_be32(o, kAlac);
    put_be32(o, 0);
    o.insert(o.end(), cookie.begin(), cookie.end());
    return o;
}

std::string parse_alac_sample_description(const Bytes &b, Bytes &cookie, uint32_t &channels, uint32_t &bits,
                                          uint32_t &sampleRate16_16)
{
    if (b.size() < 16 + 36 + 12 + 24) return "sample description too short";
    if (be32(&b[4]) != kStsd || be32(&b[0]) > b.size()) return "not a sample description box";
    if (be32(&b[12]) < 1) return "no sample entry";
    const size_t e = 16;
    const uint32_t entry = be32(&b[e]);
    if (be32(&b[e + 4]) != kAlac) return "sample entry is not 'alac'";
    if (entry < 36 + 12 + 24 || e + entry > b.size()) return "bad sample entry size";
    channels = ((uint32_t)b[e + 24] << 8) | b[e + 25];
    bits = ((uint32_t)b[e + 26] << 8) | b[e + 27];
    sampleRate16_16 = be32(&b[e + 32]);
    const size_t info = e + 36;
    const uint32_t infoSize = be32(&b[info]);
    if (be32(&b[info + 4]) != kAlac || infoSize < 12 + 24 || info + infoSize > e + entry) return "bad ALAC specific info";
    cookie.assign(b.begin() + info + 12, b.begin() + info + infoSize);
    return "";
}

// ---------------------------------------------------------------------------------------------
// ALAC in MP4 / M4A (ISO/IEC 14496-12 boxes around the sample description of ALACMagicCookieDescription.txt:177-216)
// ---------------------------------------------------------------------------------------------
namespace {

// a box under construction: the 4-byte size is patched when it is closed
struct BoxWriter {
    Bytes &o;
    std::vector<size_t> open;
    explicit BoxWriter(Bytes &out) : o(out) {}
    void begin(const char *type)
    {
        open.push_back(o.size());
        put_be32(o, 0);
        put_tag(o, type);
    }
    void full(const char *type, uint32_t versionFlags)
    {
        begin(type);
        put_be32(o, versionFlags);
    }
    void end()
    {
        const size_t at = open.back();
        open.pop_back();
        const uint32_t size = (uint32_t)(o.size() - at);
        o[at] = (uint8_t)(size >> 24);
        o[at + 1] = (uint8_t)(size >> 16);
        o[at + 2] = (uint8_t)(size >> 8);
        o[at + 3] = (uint8_t)size;
    }
};

void put_matrix(Bytes &o)
{
    const uint32_t m[9] = {0x00010000u, 0, 0, 0, 0x00010000u, 0, 0, 0, 0x40000000u};  // unity
    for (int i = 0; i < 9; i++) put_be32(o, m[i]);
}

// the movie box for one ALAC track whose single chunk starts at chunkOffset
void put_moov(Bytes &o, const AlacM4aParams &p, const Bytes &cookie, const std::vector<uint32_t> &packetBytes, uint64_t chunkOffset,
              bool wide)
{
    BoxWriter w(o);
    const uint32_t duration = (uint32_t)(p.totalFrames > 0xffffffffull ? 0xffffffffull : p.totalFrames);
    const uint32_t np = (uint32_t)packetBytes.size();
    w.begin("moov");
    w.full("mvhd", 0);
    put_be32(o, 0);  // creation / modification time
    put_be32(o, 0);
    put_be32(o, p.sampleRate);  // timescale: one tick per sample-frame
    put_be32(o, duration);
    put_be32(o, 0x00010000u);  // rate 1.0
    put_be16(o, 0x0100);       // volume 1.0
    put_zeros(o, 10);
    put_matrix(o);
    put_zeros(o, 24);          // pre_defined
    put_be32(o, 2);            // next track id
    w.end();
    w.begin("trak");
    w.full("tkhd", 7);         // enabled | in movie | in preview
    put_be32(o, 0);
    put_be32(o, 0);
    put_be32(o, 1);            // track id
    put_be32(o, 0);
    put_be32(o, duration);
    put_zeros(o, 8);
    put_be16(o, 0);            // layer
    put_be16(o, 0);            // alternate group
    put_be16(o, 0x0100);       // volume
    put_be16(o, 0);
    put_matrix(o);
    put_be32(o, 0);            // width, height
    put_be32(o, 0);
    w.end();
    w.begin("mdia");
    w.full("mdhd", 0);
    put_be32(o, 0);
    put_be32(o, 0);
    put_be32(o, p.sampleRate);
    put_be32(o, duration);
    put_be16(o, 0x55c4);       // language 'und'
    put_be16(o, 0);
    w.end();
    w.full("hdlr", 0);
    put_be32(o, 0);
    put_tag(o, "soun");
    put_zeros(o, 12);
    const char name[] = "SoundHandler";
    o.insert(o.end(), name, name + sizeof(name));  // with the terminating zero
    w.end();
    w.begin("minf");
    w.full("smhd", 0);
    put_be32(o, 0);            // balance, reserved
    w.end();
    w.begin("dinf");
    w.full("dref", 0);
    put_be32(o, 1);
    w.full("url ", 1);         // self-contained
    w.end();
    w.end();
    w.end();
    w.begin("stbl");
    {
        const Bytes stsd = build_alac_sample_description(cookie, p.channels, p.bitDepth, p.sampleRate);
        o.insert(o.end(), stsd.begin(), stsd.end());
    }
    // time to sample: full packets of framesPerPacket, then what totalFrames leaves for the last one
    w.full("stts", 0);
    {
        const uint64_t full = p.framesPerPacket ? p.totalFrames / p.framesPerPacket : 0;
        const uint32_t rest = p.framesPerPacket ? (uint32_t)(p.totalFrames - full * p.framesPerPacket) : 0;
        std::vector<std::pair<uint32_t, uint32_t> > runs;
        const uint32_t nFull = (uint32_t)(full < np ? full : np);
        if (nFull) runs.push_back(std::make_pair(nFull, p.framesPerPacket));
        if (np > nFull) runs.push_back(std::make_pair(np - nFull, rest ? rest : p.framesPerPacket));
        put_be32(o, (uint32_t)runs.size());
        for (size_t i = 0; i < runs.size(); i++) {
            put_be32(o, runs[i].first);
            put_be32(o, runs[i].second);
        }
    }
    w.end();
    w.full("stsc", 0);         // one chunk with every sample
    put_be32(o, np ? 1 : 0);
    if (np) {
        put_be32(o, 1);
        put_be32(o, np);
        put_be32(o, 1);
    }
    w.end();
    w.full("stsz", 0);
    put_be32(o, 0);            // sizes follow
    put_be32(o, np);
    for (uint32_t i = 0; i < np; i++) put_be32(o, packetBytes[i]);
    w.end();
    if (wide) {
        w.full("co64", 0);
        put_be32(o, np ? 1 : 0);
        if (np) put_be64(o, chunkOffset);
    } else {
        w.full("stco", 0);
        put_be32(o, np ? 1 : 0);
        if (np) put_be32(o, (uint32_t)chunkOffset);
    }
    w.end();
    w.end();  // stbl
    w.end();  // minf
    w.end();  // mdia
    w.end();  // trak
    w.end();  // moov
}

struct BoxRef {
    size_t body, end;  // payload range of a box
    bool ok;
};

// first child box of `type` inside [from, to)
BoxRef find_box(const Bytes &f, size_t from, size_t to, const char *type)
{
    size_t pos = from;
    while (pos + 8 <= to) {
        uint64_t size = be32(&f[pos]);
        size_t hdr = 8;
        if (size == 1) {
            if (pos + 16 > to) break;
            size = ((uint64_t)be32(&f[pos + 8]) << 32) | be32(&f[pos + 12]);
            hdr = 16;
        } else if (size == 0) {
            size = to - pos;  // "to the end of the file"
        }
        if (size < hdr || pos + size > to) break;
        if (tag_is(&f[pos + 4], type)) {
            BoxRef r = {pos + hdr, (size_t)(pos + size), true};
            return r;
        }
        pos += (size_t)size;
    }
    BoxRef none = {0, 0, false};
    return none;
}

}  // namespace

Bytes build_alac_m4a(const AlacM4aParams &p, const Bytes &cookie, const std::vector<uint32_t> &packetBytes, const uint8_t *stream,
                     uint64_t streamBytes)
{
    Bytes head;
    {
        BoxWriter w(head);
        w.begin("ftyp");
        put_tag(head, "M4A ");
        put_be32(head, 0);
        put_tag(head, "M4A ");
        put_tag(head, "mp42");
        put_tag(head, "isom");
        w.end();
    }
    // the movie box is laid out twice: its size does not depend on the chunk offset it stores, only on stco / co64
    Bytes probe;
    put_moov(probe, p, cookie, packetBytes, 0, false);
    const bool wide = head.size() + probe.size() + 16 + streamBytes > 0xffffffffull;
    if (wide) {
        probe.clear();
        put_moov(probe, p, cookie, packetBytes, 0, true);
    }
    const uint64_t mdatHeader = wide ? 16 : 8;
    const uint64_t chunkOffset = head.size() + probe.size() + mdatHeader;
    Bytes o(head);
    o.reserve((size_t)(chunkOffset + streamBytes));
    put_moov(o, p, cookie, packetBytes, chunkOffset, wide);
    if (wide) {
        put_be32(o, 1);
        put_tag(o, "mdat");
        put_be64(o, 16 + streamBytes);
    } else {
        put_be32(o, (uint32_t)(8 + streamBytes));
        put_tag(o, "mdat");
    }
    o.insert(o.end(), stream, stream + streamBytes);
    return o;
}

std::string parse_alac_m4a(const Bytes &f, InputInfo &info, AlacCafContents &out)
{
    memset(&info, 0, sizeof(info));
    out.cookie.clear();
    out.packetBytes.clear();
    out.packetPos.clear();
    out.dataPos = 0;
    info.kind = kM4aFile;
    if (f.size() < 16 || !tag_is(&f[4], "ftyp")) return "not an MP4 / M4A file";
    const BoxRef moov = find_box(f, 0, f.size(), "moov");
    if (!moov.ok) return "no moov box";
    // the first track whose sample description is 'alac'
    size_t tpos = moov.body;
    for (;;) {
        const BoxRef trak = find_box(f, tpos, moov.end, "trak");
        if (!trak.ok) return "no ALAC track";
        tpos = trak.end;
        const BoxRef mdia = find_box(f, trak.body, trak.end, "mdia");
        if (!mdia.ok) continue;
        const BoxRef minf = find_box(f, mdia.body, mdia.end, "minf");
        if (!minf.ok) continue;
        const BoxRef stbl = find_box(f, minf.body, minf.end, "stbl");
        if (!stbl.ok) continue;
        const BoxRef stsd = find_box(f, stbl.body, stbl.end, "stsd");
        if (!stsd.ok || stsd.end - stsd.body < 16 + 36 || !tag_is(&f[stsd.body + 12], "alac")) continue;
        // (the parser of the sample description wants the whole box, header included)
        const Bytes whole(f.begin() + (stsd.body - 8), f.begin() + stsd.end);
        uint32_t ch = 0, bits = 0, rate16 = 0;
        const std::string err = parse_alac_sample_description(whole, out.cookie, ch, bits, rate16);
        if (!err.empty()) return err;
        const Bytes bare = unwrap_cookie(out.cookie);
        if (bare.size() < 24) return "bad magic cookie in the sample description";
        info.isAlac = true;
        info.channels = bare[9];                       // the cookie decides (ALACSpecificConfig)
        info.sampleRate = (double)be32(&bare[20]);
        info.framesPerPacket = be32(&bare[0]);
        info.alacSourceFlag = bare[5] == 16 ? 1 : bare[5] == 20 ? 2 : bare[5] == 24 ? 3 : bare[5] == 32 ? 4 : 0;
        (void)ch;
        (void)bits;
        (void)rate16;
        // sample sizes
        const BoxRef stsz = find_box(f, stbl.body, stbl.end, "stsz");
        if (!stsz.ok || stsz.end - stsz.body < 12) return "no stsz box";
        const uint32_t fixed = be32(&f[stsz.body + 4]), count = be32(&f[stsz.body + 8]);
        if (!fixed && (uint64_t)count * 4 > stsz.end - stsz.body - 12) return "truncated stsz box";
        std::vector<uint32_t> sizes(count);
        for (uint32_t i = 0; i < count; i++) sizes[i] = fixed ? fixed : be32(&f[stsz.body + 12 + 4 * (size_t)i]);
        // chunk offsets
        std::vector<uint64_t> chunks;
        const BoxRef stco = find_box(f, stbl.body, stbl.end, "stco");
        const BoxRef co64 = find_box(f, stbl.body, stbl.end, "co64");
        if (stco.ok && stco.end - stco.body >= 8) {
            const uint32_t n = be32(&f[stco.body + 4]);
            if ((uint64_t)n * 4 > stco.end - stco.body - 8) return "truncated stco box";
            for (uint32_t i = 0; i < n; i++) chunks.push_back(be32(&f[stco.body + 8 + 4 * (size_t)i]));
        } else if (co64.ok && co64.end - co64.body >= 8) {
            const uint32_t n = be32(&f[co64.body + 4]);
            if ((uint64_t)n * 8 > co64.end - co64.body - 8) return "truncated co64 box";
            for (uint32_t i = 0; i < n; i++)
                chunks.push_back(((uint64_t)be32(&f[co64.body + 8 + 8 * (size_t)i]) << 32) | be32(&f[co64.body + 12 + 8 * (size_t)i]));
        } else if (count) {
            return "no chunk offset box";
        }
        // sample to chunk runs: (first chunk, samples per chunk, description index), 1-based
        const BoxRef stsc = find_box(f, stbl.body, stbl.end, "stsc");
        if (!stsc.ok || stsc.end - stsc.body < 8) return "no stsc box";
        const uint32_t runs = be32(&f[stsc.body + 4]);
        if ((uint64_t)runs * 12 > stsc.end - stsc.body - 8) return "truncated stsc box";
        uint32_t sample = 0;
        for (uint32_t r = 0; r < runs && sample < count; r++) {
            const uint8_t *e = &f[stsc.body + 8 + 12 * (size_t)r];
            const uint32_t first = be32(e), per = be32(e + 4);
            const uint32_t nextFirst = r + 1 < runs ? be32(e + 12) : (uint32_t)chunks.size() + 1;
            if (first < 1 || nextFirst < first) return "bad stsc box";
            for (uint32_t c = first; c < nextFirst && c <= chunks.size() && sample < count; c++) {
                uint64_t pos = chunks[c - 1];
                for (uint32_t k = 0; k < per && sample < count; k++, sample++) {
                    if (pos + sizes[sample] > f.size()) return "a packet lies outside the file";
                    out.packetPos.push_back(pos);
                    out.packetBytes.push_back(sizes[sample]);
                    pos += sizes[sample];
                }
            }
        }
        if (sample != count) return "the chunk tables do not cover every sample";
        out.dataPos = out.packetPos.empty() ? 0 : out.packetPos[0];
        info.dataPos = out.dataPos;
        uint64_t total = 0;
        for (uint32_t i = 0; i < count; i++) total += sizes[i];
        info.dataSize = total;
        return "";
    }
}

bool has_m4a_extension(const std::string &path)
{
    const size_t n = path.size();
    if (n < 4) return false;
    std::string e = path.substr(n - 4);
    for (size_t i = 0; i < e.size(); i++) e[i] = (char)tolower((unsigned char)e[i]);
    return e == ".m4a" || e == ".mp4";
}

Bytes build_pcm_caf(double sampleRate, uint32_t channels, uint32_t bits, const uint8_t *pcm, uint64_t pcmBytes)
{
    // main.cu:675-693 with SetOutputFormat's decode branch (:303-331) and WriteCAFFdescChunk's lpcm flags (:73-83)
    Bytes o;
    o.reserve((size_t)pcmBytes + 80);
    const uint32_t bytesPerFrame = channels * bytes_per_sample(bits);
    put_caf_file_header(o);
    put_desc(o, sampleRate, 0x6c70636du, 2 /* little endian */, bytesPerFrame, 1, channels, bits);
    if (channels > 2) put_chan(o, kLayoutTags[(channels - 1) & 7]);
    put_tag(o, "data");
    put_be64(o, pcmBytes + 4);
    put_be32(o, 1);
    o.insert(o.end(), pcm, pcm + pcmBytes);
    return o;
}

void swap_samples_in_place(uint8_t *pcm, uint64_t bytes, uint32_t bits)
{
    const uint32_t w = bits == 16 ? 2 : bits == 32 ? 4 : 3;  // "covers both 20 and 24" (main.cu:499)
    for (uint64_t i = 0; i + w <= bytes; i += w) {
        uint8_t *s = pcm + i;
        uint8_t t = s[0];
        s[0] = s[w - 1];
        s[w - 1] = t;
        if (w == 4) {
            t = s[1];
            s[1] = s[2];
            s[2] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
bool read_file(const std::string &path, Bytes &out)
{
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    out.clear();
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(fp);
    return true;
}

bool write_file(const std::string &path, const Bytes &data)
{
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) return false;
    const size_t n = data.empty() ? 0 : fwrite(data.data(), 1, data.size(), fp);
    const bool ok = (n == data.size()) && fclose(fp) == 0;
    return ok;
}

bool has_wav_extension(const std::string &path)
{
    const size_t dot = path.rfind('.');
    return dot != std::string::npos && path.compare(dot, std::string::npos, ".wav") == 0;
}

}  // namespace alacfile
