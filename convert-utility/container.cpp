/*
 * container.cpp — see container.h.  Every layout decision cites the reference line it reproduces.
 */
#include "container.h"

#include <cstdio>
#include <cstring>

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

uint32_t bytes_per_sample(uint32_t bits) { return bits >> 3; }  // the reference's (mBitsPerChannel >> 3), main.cu:389

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
    const uint32_t bytesPerFrame = channels * (bits >> 3);
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

Bytes build_pcm_caf(double sampleRate, uint32_t channels, uint32_t bits, const uint8_t *pcm, uint64_t pcmBytes)
{
    // main.cu:675-693 with SetOutputFormat's decode branch (:303-331) and WriteCAFFdescChunk's lpcm flags (:73-83)
    Bytes o;
    o.reserve((size_t)pcmBytes + 80);
    const uint32_t bytesPerFrame = channels * (bits >> 3);
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
