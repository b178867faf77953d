/*
 * ref_caf_adapter.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Flat extern "C" shims over the REFERENCE's own container code (convert-utility/CAFFileALAC.cpp, compiled where it
 * lies by oracle/Makefile with plain g++, no stand-ins) so that the tests can pin oracle/caf_oracle.py and the product's
 * convert-utility/container.cpp against it: chunk writers, the BER coder, BuildBasePacketTable and the chunk finders.
 * The reference functions work on FILE*; the shims run them on a temporary file and hand back the bytes, or on a file
 * holding the bytes the caller wants parsed.  Includes the reference headers from where they lie, copies nothing.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "CAFFileALAC.h" /* /root/reference/convert-utility/CAFFileALAC.h:186-201 */

namespace {

AudioFormatDescription make_format(double rate, uint32_t formatID, uint32_t flags, uint32_t bytesPerPacket,
                                   uint32_t framesPerPacket, uint32_t channels, uint32_t bits)
{
    AudioFormatDescription f;
    memset(&f, 0, sizeof(f));
    f.mSampleRate = rate;
    f.mFormatID = formatID;
    f.mFormatFlags = flags;
    f.mBytesPerPacket = bytesPerPacket;
    f.mFramesPerPacket = framesPerPacket;
    f.mBytesPerFrame = 0;
    f.mChannelsPerFrame = channels;
    f.mBitsPerChannel = bits;
    return f;
}

int64_t slurp(FILE *f, uint8_t *out, int64_t cap)
{
    fflush(f);
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n > cap) return -1;
    return (int64_t)fread(out, 1, (size_t)n, f);
}

FILE *file_of(const uint8_t *data, int64_t n)
{
    FILE *f = tmpfile();
    if (!f) return nullptr;
    fwrite(data, 1, (size_t)n, f);
    fflush(f);
    fseek(f, 0, SEEK_SET);
    return f;
}

}  // namespace

extern "C" {

/* CAFFileALAC.cpp:60-96: 'caff' file header + 'desc' chunk */
int64_t ref_caf_header(double rate, uint32_t formatID, uint32_t flags, uint32_t bytesPerPacket, uint32_t framesPerPacket,
                       uint32_t channels, uint32_t bits, uint8_t *out, int64_t cap)
{
    FILE *f = tmpfile();
    if (!f) return -1;
    WriteCAFFcaffChunk(f);
    WriteCAFFdescChunk(f, make_format(rate, formatID, flags, bytesPerPacket, framesPerPacket, channels, bits));
    const int64_t n = slurp(f, out, cap);
    fclose(f);
    return n;
}

/* :105-112 'kuki', :129-140 'chan', :142-161 'free', :98-103 'data' header, :114-127 chunk size field */
int64_t ref_caf_kuki(const uint8_t *cookie, uint32_t size, uint8_t *out, int64_t cap)
{
    FILE *f = tmpfile();
    if (!f) return -1;
    WriteCAFFkukiChunk(f, (void *)cookie, size);
    const int64_t n = slurp(f, out, cap);
    fclose(f);
    return n;
}

int64_t ref_caf_chan(uint32_t tag, uint8_t *out, int64_t cap)
{
    FILE *f = tmpfile();
    if (!f) return -1;
    WriteCAFFchanChunk(f, tag);
    const int64_t n = slurp(f, out, cap);
    fclose(f);
    return n;
}

int64_t ref_caf_free(uint32_t size, uint8_t *out, int64_t cap)
{
    FILE *f = tmpfile();
    if (!f) return -1;
    WriteCAFFfreeChunk(f, size);
    const int64_t n = slurp(f, out, cap);
    fclose(f);
    return n;
}

int64_t ref_caf_data_header(uint8_t *out, int64_t cap)
{
    FILE *f = tmpfile();
    if (!f) return -1;
    WriteCAFFdataChunk(f);
    const int64_t n = slurp(f, out, cap);
    fclose(f);
    return n;
}

int64_t ref_caf_chunk_size(int64_t numDataBytes, uint8_t *out, int64_t cap)
{
    FILE *f = tmpfile();
    if (!f) return -1;
    WriteCAFFChunkSize(f, numDataBytes);
    const int64_t n = slurp(f, out, cap);
    fclose(f);
    return n;
}

/* :163-187 'pakt' chunk header + packet table header (the function swaps the header in place) */
int64_t ref_caf_pakt_header(int64_t numPackets, int64_t numValidFrames, int32_t priming, int32_t remainder,
                            uint32_t tableSize, uint8_t *out, int64_t cap)
{
    FILE *f = tmpfile();
    if (!f) return -1;
    port_CAFPacketTableHeader h;
    memset(&h, 0, sizeof(h));
    h.mNumberPackets = numPackets;
    h.mNumberValidFrames = numValidFrames;
    h.mPrimingFrames = priming;
    h.mRemainderFrames = remainder;
    WriteCAFFpaktChunkHeader(f, &h, tableSize);
    const int64_t n = slurp(f, out, cap);
    fclose(f);
    return n;
}

/* :189-236 / :238-258 */
int32_t ref_caf_ber(int32_t value, uint8_t *buf5)
{
    int32_t n = 0;
    GetBERInteger(value, buf5, &n);
    return n;
}

uint32_t ref_caf_read_ber(const uint8_t *buf, int32_t *ioNumBytes) { return ReadBERInteger((uint8_t *)buf, ioNumBytes); }

/* :260-286: out4 = {numPackets, numValidFrames, priming, remainder}; returns the maximum packet table size */
int32_t ref_caf_base_packet_table(uint32_t bits, uint32_t channels, int32_t inputDataSize, int64_t *out4)
{
    port_CAFPacketTableHeader h;
    memset(&h, 0, sizeof(h));
    int32_t maxTable = 0;
    BuildBasePacketTable(make_format(44100.0, 0, 0, 0, 0, channels, bits), inputDataSize, &maxTable, &h);
    out4[0] = h.mNumberPackets;
    out4[1] = h.mNumberValidFrames;
    out4[2] = h.mPrimingFrames;
    out4[3] = h.mRemainderFrames;
    return maxTable;
}

/* the finders of the decode side, run on a file holding `data`:
 * out = {paktFound, paktPos, paktSize, cookieSize, dataFound, dataPos, dataSize, descFound, formatID, formatFlags,
 *        framesPerPacket, channels, bitsPerChannel, sampleRate (integer part)}; cookie receives the 'kuki' payload */
int32_t ref_caf_parse(const uint8_t *data, int64_t n, int64_t *out14, uint8_t *cookie, uint32_t cookieCap)
{
    FILE *f = file_of(data, n);
    if (!f) return -1;
    int32_t pos = 0, size = 0;
    out14[0] = FindCAFFPacketTableStart(f, &pos, &size);
    out14[1] = pos;
    out14[2] = size;
    fseek(f, 0, SEEK_SET);
    const uint32_t cs = GetMagicCookieSizeFromCAFFkuki(f);
    out14[3] = cs;
    if (cs && cs <= cookieCap) {
        uint32_t io = cs;
        fseek(f, 0, SEEK_SET);
        GetMagicCookieFromCAFFkuki(f, cookie, &io);
    }
    fseek(f, 0, SEEK_SET);
    pos = size = 0;
    out14[4] = FindCAFFDataStart(f, &pos, &size) ? 1 : 0;
    out14[5] = pos;
    out14[6] = size;
    // GetCAFFdescFormat is entered with the 4-byte file type already consumed (convert-utility/main.cu reads it to tell
    // 'caff' from 'RIFF') and has no end-of-file test: from any other position it never returns
    fseek(f, 4, SEEK_SET);
    AudioFormatDescription fmt;
    memset(&fmt, 0, sizeof(fmt));
    out14[7] = GetCAFFdescFormat(f, &fmt) ? 1 : 0;
    out14[8] = fmt.mFormatID;
    out14[9] = fmt.mFormatFlags;
    out14[10] = fmt.mFramesPerPacket;
    out14[11] = fmt.mChannelsPerFrame;
    out14[12] = fmt.mBitsPerChannel;
    out14[13] = (int64_t)fmt.mSampleRate;
    fclose(f);
    return 0;
}

}  // extern "C"
