"""CPU: the BitBuffer routines exported by libalac_hip.so (include/alac/ALACBitUtilities.h, the reference's
codec/ALACBitUtilities.h:85-97 surface) against the reference's own compiled ALACBitUtilities.c (oracle/_ref) under
random operation sequences, plus known bytes that need no reference."""
import ctypes as C

import numpy as np
import pytest

import alac_amd


class BitBuffer(C.Structure):
    _fields_ = [("cur", C.c_void_p), ("end", C.c_void_p), ("bitIndex", C.c_uint32), ("byteSize", C.c_uint32)]


def bind(lib):
    P = C.POINTER(BitBuffer)
    lib.BitBufferInit.argtypes = [P, C.c_void_p, C.c_uint32]
    lib.BitBufferInit.restype = None
    for n, r, a in (("BitBufferRead", C.c_uint32, [P, C.c_uint8]), ("BitBufferReadSmall", C.c_uint8, [P, C.c_uint8]),
                    ("BitBufferReadOne", C.c_uint8, [P]), ("BitBufferPeek", C.c_uint32, [P, C.c_uint8]),
                    ("BitBufferPeekOne", C.c_uint32, [P]), ("BitBufferUnpackBERSize", C.c_uint32, [P]),
                    ("BitBufferGetPosition", C.c_uint32, [P]), ("BitBufferByteAlign", None, [P, C.c_int32]),
                    ("BitBufferAdvance", None, [P, C.c_uint32]), ("BitBufferRewind", None, [P, C.c_uint32]),
                    ("BitBufferWrite", None, [P, C.c_uint32, C.c_uint32]), ("BitBufferReset", None, [P])):
        f = getattr(lib, n)
        f.argtypes = a
        f.restype = r
    return lib


@pytest.fixture(scope="module")
def ours():
    return bind(C.CDLL(alac_amd.LIB_PATH))


def test_known_bytes(ours):
    buf = np.zeros(16, np.uint8)
    bb = BitBuffer()
    ours.BitBufferInit(C.byref(bb), buf.ctypes.data, 12)
    ours.BitBufferWrite(C.byref(bb), 1, 3)        # ID_CPE
    ours.BitBufferWrite(C.byref(bb), 0, 4)        # element instance tag
    ours.BitBufferWrite(C.byref(bb), 0xABC, 12)
    ours.BitBufferWrite(C.byref(bb), 0xDEADBEEF, 32)
    assert ours.BitBufferGetPosition(C.byref(bb)) == 51
    ours.BitBufferByteAlign(C.byref(bb), 1)
    assert ours.BitBufferGetPosition(C.byref(bb)) == 56
    # 001 0000 1010 1011 1100 | deadbeef | 00000
    bits = "0010000" + format(0xABC, "012b") + format(0xDEADBEEF, "032b") + "00000"
    want = [int(bits[i:i + 8], 2) for i in range(0, 56, 8)]
    assert list(buf[:7]) == want
    ours.BitBufferReset(C.byref(bb))
    assert ours.BitBufferReadSmall(C.byref(bb), 3) == 1
    assert ours.BitBufferReadSmall(C.byref(bb), 4) == 0
    assert ours.BitBufferPeek(C.byref(bb), 12) == 0xABC
    assert ours.BitBufferRead(C.byref(bb), 12) == 0xABC
    assert ours.BitBufferRead(C.byref(bb), 16) == 0xDEAD
    assert ours.BitBufferPeekOne(C.byref(bb)) == 1 and ours.BitBufferReadOne(C.byref(bb)) == 1
    ours.BitBufferRewind(C.byref(bb), 17)
    assert ours.BitBufferRead(C.byref(bb), 16) == 0xDEAD
    ours.BitBufferRewind(C.byref(bb), 1000)
    assert ours.BitBufferGetPosition(C.byref(bb)) == 0
    # BER: 0x81 0x00 = 128, 0x7f = 127
    buf[:3] = [0x81, 0x00, 0x7F]
    assert ours.BitBufferUnpackBERSize(C.byref(bb)) == 128
    assert ours.BitBufferUnpackBERSize(C.byref(bb)) == 127


def test_random_sequences_match_reference_object(ours, ref):
    theirs = bind(ref.lib)
    rng = np.random.default_rng(20260410)
    for trial in range(200):
        size = int(rng.integers(8, 200))
        init = rng.integers(0, 256, size + 8, dtype=np.uint8)
        a, b = init.copy(), init.copy()
        ba, bb = BitBuffer(), BitBuffer()
        ours.BitBufferInit(C.byref(ba), a.ctypes.data, size)
        theirs.BitBufferInit(C.byref(bb), b.ctypes.data, size)
        for _ in range(int(rng.integers(5, 80))):
            pos = theirs.BitBufferGetPosition(C.byref(bb))
            assert ours.BitBufferGetPosition(C.byref(ba)) == pos
            room = size * 8 - pos
            op = int(rng.integers(0, 11))
            if op == 0 and room >= 32:
                nb = int(rng.integers(0, 33))
                val = int(rng.integers(0, 1 << 32))
                ours.BitBufferWrite(C.byref(ba), val, nb)
                theirs.BitBufferWrite(C.byref(bb), val, nb)
            elif op == 1 and room >= 16:
                nb = int(rng.integers(0, 17))
                assert ours.BitBufferRead(C.byref(ba), nb) == theirs.BitBufferRead(C.byref(bb), nb)
            elif op == 2 and room >= 8:
                nb = int(rng.integers(0, 9))
                assert ours.BitBufferReadSmall(C.byref(ba), nb) == theirs.BitBufferReadSmall(C.byref(bb), nb)
            elif op == 3 and room >= 1:
                assert ours.BitBufferReadOne(C.byref(ba)) == theirs.BitBufferReadOne(C.byref(bb))
            elif op == 4 and room >= 16:
                nb = int(rng.integers(0, 17))
                assert ours.BitBufferPeek(C.byref(ba), nb) == theirs.BitBufferPeek(C.byref(bb), nb)
                assert ours.BitBufferPeekOne(C.byref(ba)) == theirs.BitBufferPeekOne(C.byref(bb))
            elif op == 5 and room >= 8:
                z = int(rng.integers(0, 2))
                ours.BitBufferByteAlign(C.byref(ba), z)
                theirs.BitBufferByteAlign(C.byref(bb), z)
            elif op == 6 and room >= 1:
                nb = int(rng.integers(0, min(room, 70) + 1))
                ours.BitBufferAdvance(C.byref(ba), nb)
                theirs.BitBufferAdvance(C.byref(bb), nb)
            elif op == 7:
                nb = int(rng.integers(0, pos + 20))
                ours.BitBufferRewind(C.byref(ba), nb)
                theirs.BitBufferRewind(C.byref(bb), nb)
            elif op == 8 and room >= 48:
                # a BER size of up to 4 bytes written at a byte boundary, then read back
                ours.BitBufferByteAlign(C.byref(ba), 0)
                theirs.BitBufferByteAlign(C.byref(bb), 0)
                val = int(rng.integers(0, 1 << 28))
                groups = [(val >> s) & 0x7F for s in (21, 14, 7, 0)]
                while len(groups) > 1 and groups[0] == 0:
                    groups.pop(0)
                for i, g in enumerate(groups):
                    byte = g | (0x80 if i + 1 < len(groups) else 0)
                    ours.BitBufferWrite(C.byref(ba), byte, 8)
                    theirs.BitBufferWrite(C.byref(bb), byte, 8)
                ours.BitBufferRewind(C.byref(ba), 8 * len(groups))
                theirs.BitBufferRewind(C.byref(bb), 8 * len(groups))
                assert ours.BitBufferUnpackBERSize(C.byref(ba)) == theirs.BitBufferUnpackBERSize(C.byref(bb)) == val
            elif op == 9:
                ours.BitBufferReset(C.byref(ba))
                theirs.BitBufferReset(C.byref(bb))
            assert ba.bitIndex == bb.bitIndex
            assert ba.cur - a.ctypes.data == bb.cur - b.ctypes.data
            assert np.array_equal(a, b), f"trial {trial}: buffers differ after op {op}"
