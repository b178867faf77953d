"""Host-callable stage functions with the reference's prototypes (include/alac/dplib.h, aglib.h; GPU-backed one-row
batches in alac_stage_compat.cpp) against the oracle / the reference's own stage objects."""
import ctypes as C

import numpy as np
import pytest

import alac_amd

pytestmark = pytest.mark.gpu


class BitBuffer(C.Structure):
    _fields_ = [("cur", C.POINTER(C.c_uint8)), ("end", C.POINTER(C.c_uint8)), ("bitIndex", C.c_uint32), ("byteSize", C.c_uint32)]


class AGParamRec(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("mb", "mb0", "pb", "kb", "wb", "qb", "fw", "sw", "maxrun")]


@pytest.fixture(scope="module")
def lib(gpu_ctx):
    return C.CDLL(alac_amd.LIB_PATH)


def signal(n, seed, amp=3000):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    return (amp * np.sin(t * 0.031) + rng.integers(-amp // 10, amp // 10 + 1, n)).astype(np.int32)


@pytest.mark.parametrize("num,na,chanbits", [(512, 8, 17), (128, 4, 17), (300, 8, 16), (64, 16, 17), (100, 31, 17), (4096, 8, 21)])
def test_pc_unpc_block_host_prototypes(lib, oracle, num, na, chanbits):
    x = signal(num + 40, num + na, amp=3000 if chanbits < 20 else 200000)
    coefs = np.zeros(32, np.int16)
    lib.init_coefs(coefs.ctypes.data_as(C.c_void_p), 9, 16)
    assert list(coefs[:4]) == [1216, -928, -64, 0]
    want_pc, want_co = oracle.pc_block(x, num, coefs, na, chanbits)
    pc = np.zeros(num + 40, np.int32)
    co = coefs.copy()
    lib.pc_block(x.ctypes.data_as(C.c_void_p), pc.ctypes.data_as(C.c_void_p), num, co.ctypes.data_as(C.c_void_p), na, chanbits, 9)
    assert np.array_equal(pc[:num], want_pc[:num])
    if na != 31:
        assert np.array_equal(co[:na], want_co[:na])
    back = np.zeros(num + 40, np.int32)
    co2 = coefs.copy()
    lib.unpc_block(pc.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p), num, co2.ctypes.data_as(C.c_void_p), na, chanbits, 9)
    assert np.array_equal(back[:num], x[:num])


@pytest.mark.parametrize("n,bit_size,start_bit", [(512, 17, 0), (512, 17, 5), (64, 16, 3), (4096, 17, 7)])
def test_dyn_comp_decomp_host_prototypes(lib, oracle, n, bit_size, start_bit):
    rng = np.random.default_rng(n + start_bit)
    pc = (rng.standard_normal(n) * 40).astype(np.int32)
    pc[n // 3:n // 3 + 50] = 0          # a zero run
    pc[5] = 30000                       # an escape
    want_bytes, want_bits = oracle.dyn_comp(pc, bit_size, start_bit=start_bit)[:2]
    params = AGParamRec()
    lib.set_standard_ag_params(C.byref(params), n, n)
    assert (params.mb0, params.pb, params.kb, params.wb, params.maxrun) == (10, 40, 14, (1 << 14) - 1, 255)
    buf = np.zeros(len(want_bytes) + 64, np.uint8)
    bb = BitBuffer(buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.cast(buf.ctypes.data + buf.size, C.POINTER(C.c_uint8)), start_bit,
                   buf.size)
    nbits = C.c_uint32(0)
    lib.dyn_comp.restype = C.c_int32
    rc = lib.dyn_comp(C.byref(params), pc.ctypes.data_as(C.c_void_p), C.byref(bb), n, bit_size, C.byref(nbits))
    assert rc == 0 and nbits.value == want_bits
    total = start_bit + nbits.value
    assert np.array_equal(buf[:(total + 7) // 8], np.asarray(want_bytes)[:(total + 7) // 8])
    assert C.addressof(bb.cur.contents) - buf.ctypes.data == total // 8 and bb.bitIndex == total % 8
    # and back
    bb2 = BitBuffer(buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.cast(buf.ctypes.data + buf.size, C.POINTER(C.c_uint8)), start_bit,
                    buf.size)
    out = np.zeros(n, np.int32)
    used = C.c_uint32(0)
    lib.dyn_decomp.restype = C.c_int32
    rc = lib.dyn_decomp(C.byref(params), C.byref(bb2), out.ctypes.data_as(C.c_void_p), n, bit_size, C.byref(used))
    assert rc == 0 and used.value == nbits.value
    assert np.array_equal(out, pc)
