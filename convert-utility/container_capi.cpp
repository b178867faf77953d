/*
 * container_capi.cpp — C entry points over container.cpp so that the CPU-side tests can drive the container
 * code through ctypes without a GPU (tests/test_container.py).  Not part of the product binary.
 */
#include <cstring>

#include "container.h"

using namespace alacfile;

extern "C" {

struct alacfile_info {
    int32_t kind, is_alac, big_endian_pcm;
    double sample_rate;
    uint32_t channels, bits_per_channel, alac_source_flag, frames_per_packet;
    uint64_t data_pos, data_size;
};

/* returns 0 on success, -1 with the diagnostic in err[errcap] */
int32_t alacfile_sniff(const uint8_t *file, uint64_t size, alacfile_info *out, char *err, uint32_t errcap)
{
    Bytes f(file, file + size);
    InputInfo info;
    const std::string e = sniff_input(f, info);
    if (err && errcap) {
        strncpy(err, e.c_str(), errcap - 1);
        err[errcap - 1] = 0;
    }
    out->kind = info.kind;
    out->is_alac = info.isAlac;
    out->big_endian_pcm = info.bigEndianPcm;
    out->sample_rate = info.sampleRate;
    out->channels = info.channels;
    out->bits_per_channel = info.bitsPerChannel;
    out->alac_source_flag = info.alacSourceFlag;
    out->frames_per_packet = info.framesPerPacket;
    out->data_pos = info.dataPos;
    out->data_size = info.dataSize;
    return e.empty() ? 0 : -1;
}

static uint64_t give(const Bytes &b, uint8_t *out, uint64_t cap)
{
    if (out && !b.empty() && b.size() <= cap) memcpy(out, b.data(), b.size());
    return b.size();
}

/* each builder returns the size of the file image; it is copied to out when it fits in cap */
uint64_t alacfile_build_alac_caf(double sample_rate, uint32_t channels, uint32_t bit_depth, uint32_t frames_per_packet,
                                 uint64_t input_data_bytes, const uint8_t *cookie, uint32_t cookie_size,
                                 const uint32_t *packet_bytes, uint32_t num_packets, const uint8_t *stream,
                                 uint64_t stream_bytes, uint8_t *out, uint64_t cap)
{
    AlacCafParams p = {sample_rate, channels, bit_depth, frames_per_packet, input_data_bytes};
    Bytes ck(cookie, cookie + cookie_size);
    std::vector<uint32_t> sizes(packet_bytes, packet_bytes + num_packets);
    return give(build_alac_caf(p, ck, sizes, stream, stream_bytes), out, cap);
}

uint64_t alacfile_build_wave(double sample_rate, uint32_t channels, uint32_t bits, const uint8_t *pcm, uint64_t pcm_bytes,
                             uint8_t *out, uint64_t cap)
{
    return give(build_wave(sample_rate, channels, bits, pcm, pcm_bytes), out, cap);
}

uint64_t alacfile_build_pcm_caf(double sample_rate, uint32_t channels, uint32_t bits, const uint8_t *pcm, uint64_t pcm_bytes,
                                uint8_t *out, uint64_t cap)
{
    return give(build_pcm_caf(sample_rate, channels, bits, pcm, pcm_bytes), out, cap);
}

/* cookie wrappers / MP4 sample description (ALACMagicCookieDescription.txt:177-238); kind 0 = wrap legacy, 1 = unwrap */
uint64_t alacfile_cookie(int32_t kind, const uint8_t *cookie, uint32_t size, uint8_t *out, uint64_t cap)
{
    Bytes c(cookie, cookie + size);
    return give(kind == 0 ? wrap_legacy_cookie(c) : unwrap_cookie(c), out, cap);
}

uint64_t alacfile_build_stsd(const uint8_t *cookie, uint32_t size, uint32_t channels, uint32_t bits, uint32_t rate,
                             uint8_t *out, uint64_t cap)
{
    return give(build_alac_sample_description(Bytes(cookie, cookie + size), channels, bits, rate), out, cap);
}

/* returns the cookie size (copied to cookie_out, cap 64) or -1; fields3 = channels, bits, rate 16.16 */
int64_t alacfile_parse_stsd(const uint8_t *box, uint64_t size, uint8_t *cookie_out, uint32_t *fields3)
{
    Bytes ck;
    if (!parse_alac_sample_description(Bytes(box, box + size), ck, fields3[0], fields3[1], fields3[2]).empty() || ck.size() > 64)
        return -1;
    if (!ck.empty()) memcpy(cookie_out, ck.data(), ck.size());
    return (int64_t)ck.size();
}

/* cookie to cookie_out (cap 64), packet sizes to sizes_out (cap max_packets); returns the packet count or -1 */
int64_t alacfile_parse_alac_caf(const uint8_t *file, uint64_t size, uint8_t *cookie_out, uint32_t *cookie_size,
                                uint32_t *sizes_out, uint32_t max_packets, uint64_t *data_pos)
{
    Bytes f(file, file + size);
    InputInfo info;
    if (!sniff_input(f, info).empty() || !info.isAlac) return -1;
    AlacCafContents c;
    if (!parse_alac_caf(f, info, c).empty()) return -1;
    *cookie_size = (uint32_t)c.cookie.size();
    if (!c.cookie.empty() && c.cookie.size() <= 64) memcpy(cookie_out, c.cookie.data(), c.cookie.size());
    for (size_t i = 0; i < c.packetBytes.size() && i < max_packets; i++) sizes_out[i] = c.packetBytes[i];
    *data_pos = c.dataPos;
    return (int64_t)c.packetBytes.size();
}

/* MP4 / M4A: the file image of one ALAC track */
uint64_t alacfile_build_alac_m4a(uint32_t sample_rate, uint32_t channels, uint32_t bit_depth, uint32_t frames_per_packet,
                                 uint64_t total_frames, const uint8_t *cookie, uint32_t cookie_size, const uint32_t *packet_bytes,
                                 uint32_t num_packets, const uint8_t *stream, uint64_t stream_bytes, uint8_t *out, uint64_t cap)
{
    AlacM4aParams p = {sample_rate, channels, bit_depth, frames_per_packet, total_frames};
    Bytes ck(cookie, cookie + cookie_size);
    std::vector<uint32_t> sizes(packet_bytes, packet_bytes + num_packets);
    return give(build_alac_m4a(p, ck, sizes, stream, stream_bytes), out, cap);
}

/* cookie to cookie_out (cap 64), packet sizes / file offsets to sizes_out / pos_out (cap max_packets); returns the packet count,
 * or -1 with the diagnostic in err */
int64_t alacfile_parse_alac_m4a(const uint8_t *file, uint64_t size, alacfile_info *info_out, uint8_t *cookie_out,
                                uint32_t *cookie_size, uint32_t *sizes_out, uint64_t *pos_out, uint32_t max_packets, char *err,
                                uint32_t errcap)
{
    Bytes f(file, file + size);
    InputInfo info;
    AlacCafContents c;
    const std::string e = parse_alac_m4a(f, info, c);
    if (err && errcap) {
        strncpy(err, e.c_str(), errcap - 1);
        err[errcap - 1] = 0;
    }
    if (!e.empty()) return -1;
    info_out->kind = info.kind;
    info_out->is_alac = info.isAlac;
    info_out->big_endian_pcm = 0;
    info_out->sample_rate = info.sampleRate;
    info_out->channels = info.channels;
    info_out->bits_per_channel = info.bitsPerChannel;
    info_out->alac_source_flag = info.alacSourceFlag;
    info_out->frames_per_packet = info.framesPerPacket;
    info_out->data_pos = info.dataPos;
    info_out->data_size = info.dataSize;
    *cookie_size = (uint32_t)c.cookie.size();
    if (!c.cookie.empty() && c.cookie.size() <= 64) memcpy(cookie_out, c.cookie.data(), c.cookie.size());
    for (size_t i = 0; i < c.packetBytes.size() && i < max_packets; i++) {
        sizes_out[i] = c.packetBytes[i];
        pos_out[i] = c.packetPos[i];
    }
    return (int64_t)c.packetBytes.size();
}

void alacfile_swap_samples(uint8_t *pcm, uint64_t bytes, uint32_t bits) { swap_samples_in_place(pcm, bytes, bits); }

uint32_t alacfile_append_ber(uint32_t value, uint8_t *out5)
{
    Bytes b;
    append_ber(b, value);
    memcpy(out5, b.data(), b.size());
    return (uint32_t)b.size();
}

uint32_t alacfile_read_ber(const uint8_t *p, uint32_t avail, uint32_t *used)
{
    size_t u = 0;
    const uint32_t v = read_ber(p, avail, &u);
    *used = (uint32_t)u;
    return v;
}

}  // extern "C"
