"""CPU: the MP4 / M4A reader and writer of convert-utility/container.cpp.  The reference ships no MP4 code —
only the field list of the sample description (ALACMagicCookieDescription.txt:177-216) — so the checker is an independent box
walker written here from ISO/IEC 14496-12's box grammar: write -> walk -> same cookie / packets / offsets, and write -> read
-> identical packets, cookie and sample counts; plus files laid out differently from this writer's (several chunks, co64, a
fixed sample size, 'mdat' in front of 'moov', a second non-ALAC track)."""
import struct

import numpy as np
import pytest

from container_lib import Container, music_like


@pytest.fixture(scope="module")
def ct():
    return Container()


def walk(data, start=0, end=None, path=""):
    """every box of an ISO base media file -> {path: (payload start, payload end)}; containers are descended into"""
    end = len(data) if end is None else end
    out, pos = {}, start
    while pos + 8 <= end:
        size, typ = struct.unpack(">I4s", data[pos:pos + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", data[pos + 8:pos + 16])[0]
            hdr = 16
        elif size == 0:
            size = end - pos
        assert size >= hdr and pos + size <= end, (path, typ, size)
        name = path + "/" + typ.decode("latin1")
        out.setdefault(name, (pos + hdr, pos + size))
        if typ in (b"moov", b"trak", b"mdia", b"minf", b"stbl", b"dinf"):
            out.update({k: v for k, v in walk(data, pos + hdr, pos + size, name).items() if k not in out})
        pos += size
    assert pos == end, (path, pos, end)
    return out


def encode(oracle, bits, ch, frames, seed, rate=44100):
    pcm = music_like(frames, ch, bits, seed)
    enc = oracle.encoder(4096, bits, ch, rate)
    stream, sizes = enc.encode_stream(np.frombuffer(pcm, np.uint8), frames, 0)
    return pcm, enc.cookie().tobytes(), stream.tobytes(), sizes


@pytest.mark.parametrize("bits,ch,frames", [(16, 2, 4096 * 3 + 77), (16, 1, 4096 * 2), (24, 2, 5000), (16, 6, 4096 + 9), (16, 2, 1)])
def test_write_walk_read(ct, oracle, bits, ch, frames):
    pcm, cookie, stream, sizes = encode(oracle, bits, ch, frames, 3)
    f = ct.build_alac_m4a(44100, ch, bits, frames, cookie, sizes, stream)
    b = walk(f)
    for need in ("/ftyp", "/moov/mvhd", "/moov/trak/tkhd", "/moov/trak/mdia/mdhd", "/moov/trak/mdia/hdlr",
                 "/moov/trak/mdia/minf/smhd", "/moov/trak/mdia/minf/dinf/dref", "/moov/trak/mdia/minf/stbl/stsd",
                 "/moov/trak/mdia/minf/stbl/stts", "/moov/trak/mdia/minf/stbl/stsc", "/moov/trak/mdia/minf/stbl/stsz",
                 "/moov/trak/mdia/minf/stbl/stco", "/mdat"):
        assert need in b, need
    assert f[8:12] == b"M4A "
    # the independent reading of the tables
    a, e = b["/moov/trak/mdia/minf/stbl/stsz"]
    fixed, count = struct.unpack(">II", f[a + 4:a + 12])
    assert fixed == 0 and count == len(sizes)
    assert list(struct.unpack(f">{count}I", f[a + 12:a + 12 + 4 * count])) == [int(x) for x in sizes]
    a, e = b["/moov/trak/mdia/minf/stbl/stco"]
    assert struct.unpack(">II", f[a + 4:a + 12]) == (1, b["/mdat"][0])
    assert f[b["/mdat"][0]:b["/mdat"][1]] == stream
    a, e = b["/moov/trak/mdia/minf/stbl/stts"]
    n = struct.unpack(">I", f[a + 4:a + 8])[0]
    runs = [struct.unpack(">II", f[a + 8 + 8 * i:a + 16 + 8 * i]) for i in range(n)]
    assert sum(c * d for c, d in runs) == frames and sum(c for c, _ in runs) == len(sizes)
    a, e = b["/moov/trak/mdia/mdhd"]
    assert struct.unpack(">II", f[a + 12:a + 20]) == (44100, frames)
    a, e = b["/moov/trak/mdia/minf/stbl/stsd"]
    assert f[a + 12:a + 16] == b"alac" and f[a - 8:e] == ct.build_stsd(cookie, ch, bits, 44100)
    assert f[e - len(cookie):e] == cookie
    # the product's reader
    info, ck, sz, pos = ct.parse_alac_m4a(f)
    assert ck == cookie and list(sz) == [int(x) for x in sizes]
    assert list(pos) == list(b["/mdat"][0] + np.concatenate([[0], np.cumsum(sizes[:-1].astype(np.int64))]))
    assert (info.kind, info.is_alac, info.channels, int(info.sample_rate), info.frames_per_packet) == (3, 1, ch, 44100, 4096)
    assert info.alac_source_flag == {16: 1, 20: 2, 24: 3, 32: 4}[bits]
    rc, sinfo, err = ct.sniff(f)
    assert rc == 0 and sinfo.kind == 3 and sinfo.is_alac == 1, err
    # the packets decode back to the input (oracle decoder)
    dec = oracle.decoder(np.frombuffer(ck, np.uint8))
    bpf = ch * {16: 2, 24: 3}[bits]
    back = b""
    for p, s in zip(pos, sz):
        st, out, ns = dec.decode_packet(np.frombuffer(f[int(p):int(p) + int(s)], np.uint8), bpf)
        assert st == 0
        back += out.tobytes()
    assert back == pcm


def box(typ, payload):
    return struct.pack(">I4s", 8 + len(payload), typ) + payload


def full(typ, payload, vf=0):
    return box(typ, struct.pack(">I", vf) + payload)


def test_reads_other_muxers_layouts(ct, oracle):
    """three chunks of 2 + 2 + 1 packets with padding between them, 64-bit chunk offsets, 'mdat' in FRONT of 'moov', a video
    track in front of the ALAC one"""
    pcm, cookie, stream, sizes = encode(oracle, 16, 2, 4096 * 4 + 500, 9)
    assert len(sizes) == 5
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    chunks = [stream[offs[0]:offs[2]], stream[offs[2]:offs[4]], stream[offs[4]:offs[5]]]
    ftyp = box(b"ftyp", b"isom\0\0\2\0isomiso2mp41")
    mdat_payload, chunk_pos, at = b"", [], len(ftyp) + 8
    for i, c in enumerate(chunks):
        pad = b"\xee" * (13 * i)
        mdat_payload += pad + c
        chunk_pos.append(at + len(pad))
        at += len(pad) + len(c)
    mdat = box(b"mdat", mdat_payload)
    stsd = ct.build_stsd(cookie, 2, 16, 44100)
    stbl = box(b"stbl", stsd + full(b"stts", struct.pack(">III", 1, 5, 4096))
               + full(b"stsc", struct.pack(">I", 2) + struct.pack(">III", 1, 2, 1) + struct.pack(">III", 3, 1, 1))
               + full(b"stsz", struct.pack(">II", 0, 5) + struct.pack(">5I", *[int(x) for x in sizes]))
               + full(b"co64", struct.pack(">I", 3) + struct.pack(">3Q", *chunk_pos)))
    audio = box(b"trak", full(b"tkhd", b"\0" * 80, 7) + box(b"mdia", full(b"mdhd", b"\0" * 20) + full(b"hdlr", b"\0" * 4 + b"soun" + b"\0" * 13)
                                                                 + box(b"minf", full(b"smhd", b"\0" * 4) + stbl)))
    vstbl = box(b"stbl", full(b"stsd", struct.pack(">I", 1) + struct.pack(">I4s", 16, b"avc1") + b"\0" * 8))
    video = box(b"trak", full(b"tkhd", b"\0" * 80, 7) + box(b"mdia", full(b"hdlr", b"\0" * 4 + b"vide" + b"\0" * 13) + box(b"minf", vstbl)))
    f = ftyp + mdat + box(b"moov", full(b"mvhd", b"\0" * 96) + video + audio)
    info, ck, sz, pos = ct.parse_alac_m4a(f)
    assert ck == cookie and list(sz) == [int(x) for x in sizes]
    want_pos = [chunk_pos[0], chunk_pos[0] + int(sizes[0]), chunk_pos[1], chunk_pos[1] + int(sizes[2]), chunk_pos[2]]
    assert [int(x) for x in pos] == want_pos
    for p, s, a in zip(pos, sz, offs):
        assert f[int(p):int(p) + int(s)] == stream[a:a + int(s)]


def test_fixed_sample_size_and_refusals(ct, oracle):
    pcm, cookie, stream, sizes = encode(oracle, 16, 2, 4096, 5)
    f = ct.build_alac_m4a(44100, 2, 16, 4096, cookie, sizes, stream)
    assert isinstance(ct.parse_alac_m4a(f[:40]), str)                       # no moov
    bad = bytearray(f)
    i = bad.index(b"stsz")
    bad[i + 12:i + 16] = struct.pack(">I", 99)                              # more samples than the chunk tables cover
    assert isinstance(ct.parse_alac_m4a(bytes(bad)), str)
    bad = bytearray(f)
    i = bad.index(b"stco")
    bad[i + 12:i + 16] = struct.pack(">I", len(f) - 3)                      # the packet would run past the end of the file
    assert "outside the file" in ct.parse_alac_m4a(bytes(bad))
    bad = bytearray(f)
    i = bad.index(b"stsd")
    bad[i + 16:i + 20] = b"mp4a"                                            # not an ALAC track
    assert "no ALAC track" in ct.parse_alac_m4a(bytes(bad))
    # a fixed sample size (stsz.sample_size != 0) is legal for constant-size packets
    good = bytearray(f)
    i = good.index(b"stsz")
    assert struct.unpack(">I", good[i + 12:i + 16])[0] == 1
    good[i + 8:i + 12] = struct.pack(">I", int(sizes[0]))
    info, ck, sz, pos = ct.parse_alac_m4a(bytes(good))
    assert list(sz) == [int(sizes[0])]
    assert ct.sniff(b"\0\0\0\x10ftypM4A " + b"\0" * 8)[0] == -1
