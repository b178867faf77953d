"""caf_oracle.py — TEST INFRASTRUCTURE ONLY (never imported by the product).

CPU restatement of what the reference's convert utility does to FILES, written the way the reference does it:
a seekable file object that is written, sought and patched in the same order as convert-utility/main.cu and
convert-utility/CAFFileALAC.cpp (file:line cited per function).  The product (convert-utility/container.cpp)
computes the same layout in one pass without seeking; the tests compare the two byte for byte.

The codec itself is injected: `encode_packet(pcm_bytes, num_frames) -> bytes` (chained, e.g. the C oracle's
encoder) and `decode_packet(packet_bytes) -> (pcm_bytes, num_frames)`.

Parity pin: the reference's alacconvert cannot be built here (its main.cu / ALACEncoder.cu need CUDA), but its container
code can: oracle/Makefile compiles convert-utility/CAFFileALAC.cpp where it lies into oracle/_ref/libcafref.so, and
tests/test_container_refpin.py checks every chunk writer, the BER coder, base_packet_table and whole product-written
files against it (and against the committed fixture tests/golden/caf_headers.json where /root/reference is absent).
The seek-and-patch SEQUENCE of main.cu (which chunk when, the free-chunk fix-up) is restated and pinned by the known
answers recorded in SURVEY.md §8c/§8f (table entry widths, the phantom packet, 50.wav -> 237 packets / 1 164 578 payload
bytes) and by round trips.
"""
import struct

K_FRAMES = 4096          # kALACDefaultFramesPerPacket, codec/ALACAudioTypes.h:74
K_ESCAPE = 8             # kALACMaxEscapeHeaderBytes, codec/ALACAudioTypes.h:71
K_PAKT_HDR = 24          # kMinCAFFPacketTableHeaderSize, CAFFileALAC.h:32
LAYOUT_TAGS = [(100 << 16) | 1, (101 << 16) | 2, (113 << 16) | 3, (116 << 16) | 4,
               (120 << 16) | 5, (124 << 16) | 6, (142 << 16) | 7, (127 << 16) | 8]


class SeekFile:
    """fopen("w+b") semantics over a bytearray: writes past the end extend, writes inside overwrite."""

    def __init__(self, data=b""):
        self.b = bytearray(data)
        self.pos = 0

    def write(self, data):
        data = bytes(data)
        end = self.pos + len(data)
        if end > len(self.b):
            self.b.extend(b"\0" * (end - len(self.b)))
        self.b[self.pos:end] = data
        self.pos = end

    def read(self, n):
        d = bytes(self.b[self.pos:self.pos + n])
        self.pos += len(d)
        return d

    def seek(self, pos):
        self.pos = pos

    def skip(self, n):
        self.pos += n

    def tell(self):
        return self.pos


def be32(b, o=0):
    return int.from_bytes(b[o:o + 4], "big")


def le32(b, o=0):
    return int.from_bytes(b[o:o + 4], "little")


# ---------------------------------------------------------------------------------------------------
# input sniffing: GetInputFormat main.cu:196-261, GetCAFFdescFormat CAFFileALAC.cpp:395-456,
# FindDataStart main.cu:333-385, FindCAFFDataStart CAFFileALAC.cpp:363-393
# ---------------------------------------------------------------------------------------------------
def get_input_format(data):
    f = SeekFile(data)
    head = f.read(4)
    fmt = {}
    if head == b"caff":
        fmt["file"] = "caff"
        f.skip(4)
        while True:
            t = f.read(4)
            if len(t) < 4:
                return None
            if t == b"desc":
                f.skip(8)
                d = f.read(32)
                fmt["rate"] = struct.unpack(">d", d[0:8])[0]
                fmt["id"] = d[8:12]
                flags = be32(d, 12)
                fmt["bytes_per_packet"] = be32(d, 16)
                fmt["frames_per_packet"] = be32(d, 20)
                fmt["channels"] = be32(d, 24)
                fmt["bits"] = be32(d, 28)
                if fmt["id"] == b"alac":
                    fmt["flags"] = flags
                else:
                    # :428-436: CAF says "little endian" with bit 1; the in-memory flag means "big endian"
                    fmt["flags"] = (flags & 0xfffffffc) if (flags & 2) == 2 else (flags | 2)
                return fmt
            sz = f.read(8)
            f.skip(be32(sz, 4))
    if head == b"RIFF":
        rest = f.read(8)
        if rest[4:8] != b"WAVE":
            return None
        fmt["file"] = "WAVE"
        while True:
            t = f.read(4)
            if len(t) < 4:
                return None
            if t == b"fmt ":
                b = f.read(20)
                if b[4] != 1 or b[5] != 0:
                    return None
                fmt["id"] = b"lpcm"
                fmt["channels"] = b[6]
                fmt["rate"] = float(le32(b, 8))
                fmt["bits"] = b[18]
                fmt["flags"] = 0x4 | 0x8          # signed | packed, little endian
                fmt["bytes_per_packet"] = (fmt["bits"] >> 3) * fmt["channels"]
                fmt["frames_per_packet"] = 1
                return fmt
            f.skip(le32(f.read(4)))
    return None


def find_data_start(data, file_type):
    f = SeekFile(data)
    if file_type == "WAVE":
        file_size = le32(data, 4)
        f.seek(12)
        while f.tell() < file_size:
            h = f.read(8)
            if len(h) < 8:
                return None
            if h[0:4] == b"data":
                return f.tell(), le32(h, 4)
            f.skip(le32(h, 4))
        return None
    f.seek(8)
    while True:
        h = f.read(12)
        if len(h) < 12:
            return None
        if h[0:4] == b"data":
            return f.tell() + 4, be32(h, 8) - 4
        f.skip(be32(h, 8))


# ---------------------------------------------------------------------------------------------------
# chunk writers, CAFFileALAC.cpp:60-187
# ---------------------------------------------------------------------------------------------------
def write_caff(f):
    f.write(b"caff\x00\x01\x00\x00")


def write_desc(f, rate, fmt_id, flags, bytes_per_packet, frames_per_packet, channels, bits):
    f.write(b"desc" + b"\0" * 7 + bytes([32]))
    f.write(struct.pack(">d4sIIIII", rate, fmt_id, flags, bytes_per_packet, frames_per_packet, channels, bits))


def write_kuki(f, cookie):
    f.write(b"kuki" + b"\0" * 7 + bytes([len(cookie) & 0xff]))
    f.write(cookie)


def write_chan(f, tag):
    f.write(b"chan" + b"\0" * 7 + bytes([12]) + struct.pack(">I", tag) + b"\0" * 8)


def write_data_header(f):
    f.write(b"data" + b"\0" * 8 + b"\x00\x00\x00\x01")


def write_chunk_size(f, n):
    f.write(struct.pack(">q", n))


def write_free(f, size):
    adj = (size - 12) & 0xffffffff
    if size > adj:
        f.write(b"free" + b"\0" * 4 + struct.pack(">I", adj))
        f.write(b"\0" * adj)


def ber(v):
    """GetBERInteger, CAFFileALAC.cpp:189-236"""
    if (v & 0x7f) == v:
        return bytes([v])
    if (v & 0x3fff) == v:
        return bytes([(v >> 7) | 0x80, v & 0x7f])
    if (v & 0x1fffff) == v:
        return bytes([(v >> 14) | 0x80, ((v >> 7) & 0x7f) | 0x80, v & 0x7f])
    if (v & 0x0fffffff) == v:
        return bytes([(v >> 21) | 0x80, ((v >> 14) & 0x7f) | 0x80, ((v >> 7) & 0x7f) | 0x80, v & 0x7f])
    return bytes([((v >> 28) & 0xff) | 0x80, ((v >> 21) & 0x7f) | 0x80, ((v >> 14) & 0x7f) | 0x80,
                  ((v >> 7) & 0x7f) | 0x80, v & 0x7f])


def read_ber(buf, num_bytes):
    """ReadBERInteger, CAFFileALAC.cpp:238-260 -> (value, bytes used)"""
    ans, size = 0, 0
    while True:
        if size >= len(buf):
            return 0, 0
        d = buf[size]
        ans = (ans << 7) | (d & 0x7f)
        size += 1
        if size > 5:
            return 0, 0
        if not ((d & 0x80) != 0 and size <= num_bytes):
            break
    return ans, size


def base_packet_table(bits, channels, input_data_size):
    """BuildBasePacketTable, CAFFileALAC.cpp:262-287 -> (table bytes, packets, valid frames, remainder)"""
    valid = input_data_size // ((bits >> 3) * channels)
    packets = valid // K_FRAMES
    remainder = K_FRAMES - (valid - packets * K_FRAMES)
    if remainder:
        packets += 1
    max_packet = (bits >> 3) * channels * K_FRAMES + K_ESCAPE
    entry = 2 if max_packet < 16384 else 3
    return entry * packets, packets, valid, remainder


def swap_to_little(buf, bits):
    """main.cu:482-507"""
    b = bytearray(buf)
    w = 2 if bits == 16 else 4 if bits == 32 else 3
    for i in range(0, len(b) - w + 1, w):
        b[i:i + w] = b[i:i + w][::-1]
    return bytes(b)


# ---------------------------------------------------------------------------------------------------
# EncodeALAC, main.cu:387-643
# ---------------------------------------------------------------------------------------------------
def encode_file(data, cookie, encode_packet):
    """data: WAV or PCM-CAF file image; cookie: the encoder's magic cookie; encode_packet(pcm, frames) -> bytes
    (chained from call to call).  Returns the CAF file image the reference would leave on disk."""
    fmt = get_input_format(data)
    assert fmt is not None and fmt["id"] == b"lpcm"
    pos, size = find_data_start(data, fmt["file"])
    size = min(size, len(data) - pos)
    bits, ch = fmt["bits"], fmt["channels"]
    in_packet_bytes = ch * (bits >> 3) * K_FRAMES
    out = SeekFile()
    write_caff(out)
    flag = {16: 1, 20: 2, 24: 3, 32: 4}[bits]
    write_desc(out, fmt["rate"], b"alac", flag, 0, K_FRAMES, ch, 0)
    write_kuki(out, bytes(cookie))
    if ch > 2:
        write_chan(out, LAYOUT_TAGS[ch - 1])
    table_size, packets, valid, remainder = base_packet_table(bits, ch, size)
    table_left = table_size
    # WriteCAFFpaktChunkHeader (:163-187): low 32 bits of the size only
    out.write(b"pakt" + b"\0" * 4 + struct.pack(">I", (table_size + K_PAKT_HDR) & 0xffffffff))
    out.write(struct.pack(">qqii", packets, valid, 0, remainder))
    table_pos = out.tell()
    table_size_pos = table_pos - (8 + K_PAKT_HDR)
    out.write(b"\0" * table_size)
    data_size_pos = out.tell() + 4
    write_data_header(out)
    data_pos = out.tell()

    remaining, src, num_data_bytes = size, pos, 0
    while remaining > 0:
        take = in_packet_bytes if in_packet_bytes <= remaining else remaining
        pcm = data[src:src + take]
        if (fmt["flags"] & 2) != 0:
            pcm = swap_to_little(pcm, bits)
        frames = take // fmt["bytes_per_packet"]            # ALACEncoder.cu:984
        pkt = bytes(encode_packet(pcm, frames))
        b = ber(len(pkt))
        out.seek(table_pos)
        out.write(b)
        table_pos += len(b)
        table_left = (table_left - len(b)) & 0xffffffff      # uint32_t in the reference
        out.seek(data_pos)
        out.write(pkt)
        data_pos += len(pkt)
        num_data_bytes += len(pkt)
        src += take
        remaining -= take
    if table_left > 12:
        out.seek(table_pos)
        write_free(out, table_left)
        out.seek(table_size_pos)
        write_chunk_size(out, table_size - table_left + K_PAKT_HDR)
    out.seek(data_size_pos)
    write_chunk_size(out, num_data_bytes + 4)
    return bytes(out.b)


# ---------------------------------------------------------------------------------------------------
# DecodeALAC, main.cu:646-790
# ---------------------------------------------------------------------------------------------------
def find_chunk(data, tag):
    f = SeekFile(data)
    f.seek(8)
    while True:
        h = f.read(12)
        if len(h) < 12:
            return None
        if h[0:4] == tag:
            return f.tell(), h
        f.skip(be32(h, 8))


def get_cookie(data):
    pos, h = find_chunk(data, b"kuki")
    return bytes(data[pos:pos + h[11]])


def decode_file(data, to_wave, decode_packet):
    """data: ALAC CAF file image; decode_packet(packet) -> (pcm bytes, frames).  Returns the WAV / CAF image."""
    fmt = get_input_format(data)
    assert fmt is not None and fmt["id"] == b"alac"
    in_pos, in_size = find_data_start(data, "caff")
    bits = {1: 16, 2: 20, 3: 24, 4: 32}[fmt["flags"]]
    ch = fmt["channels"]
    bytes_per_frame = ch * (bits >> 3)
    out = SeekFile()
    if not to_wave:
        write_caff(out)
        write_desc(out, fmt["rate"], b"lpcm", 2, bytes_per_frame, 1, ch, bits)
        if ch > 2:
            write_chan(out, LAYOUT_TAGS[ch - 1])
        size_pos = out.tell() + 4
        write_data_header(out)
    else:
        out.write(b"RIFF\0\0\0\0WAVE")
        rate = int(fmt["rate"])
        out.write(b"fmt " + struct.pack("<IHBBIIBBBB", 16, 1, ch & 0xff, 0, rate, (rate * bytes_per_frame) & 0xffffffff,
                                        bytes_per_frame & 0xff, 0, bits & 0xff, 0))
        out.write(b"data\0\0\0\0")
        size_pos = out.tell() - 4
    tpos, _ = find_chunk(data, b"pakt")
    tpos += K_PAKT_HDR
    window = data[tpos:tpos + 5]
    size, used = read_ber(window, len(window))
    tpos += used
    dpos = in_pos
    total = 0
    while size > 0 and dpos + size <= len(data):
        pcm, frames = decode_packet(bytes(data[dpos:dpos + size]))
        nbytes = frames * bytes_per_frame
        out.write(bytes(pcm[:nbytes]))
        total += nbytes
        dpos += size
        window = data[tpos:tpos + 5]
        size, used = read_ber(window, len(window))
        tpos += used
    if not to_wave:
        out.seek(size_pos)
        write_chunk_size(out, total + 4)
    else:
        out.seek(size_pos)
        out.write(struct.pack("<I", total & 0xffffffff))
        out.seek(4)
        out.write(struct.pack("<I", (total + 4 + 8 + 24) & 0xffffffff))
    return bytes(out.b)


# ---------------------------------------------------------------------------------------------------
# helpers for tests: make input files
# ---------------------------------------------------------------------------------------------------
def make_wav(pcm, channels, rate, bits, extra_chunks=()):
    body = b"WAVE"
    for tag, payload in extra_chunks:
        body += tag + struct.pack("<I", len(payload)) + payload
    bpf = channels * (bits >> 3)
    body += b"fmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * bpf, bpf, bits)
    body += b"data" + struct.pack("<I", len(pcm)) + bytes(pcm)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def make_pcm_caf(pcm, channels, rate, bits, little_endian=True):
    bpf = channels * (bits >> 3)
    out = b"caff\x00\x01\x00\x00"
    out += b"desc" + struct.pack(">q", 32) + struct.pack(">d4sIIIII", float(rate), b"lpcm", 2 if little_endian else 0, bpf, 1,
                                                         channels, bits)
    out += b"data" + struct.pack(">q", len(pcm) + 4) + b"\x00\x00\x00\x00" + bytes(pcm)
    return out
