"""mix16 / mix20 / mix24 / mix32 / copy20ToPredictor with the reference's prototypes (include/alac/matrixlib.h,
codec/matrixlib.h:41-60; GPU kernels in alac_matrix.hip) against the oracle's restatement of codec/matrix_enc.cu,
with host buffers (Apple's convention) and device buffers (the fork's convention)."""
import ctypes as C

import numpy as np
import pytest
import torch

import alac_amd

pytestmark = pytest.mark.gpu

BPS = {16: 2, 20: 3, 24: 3, 32: 4}


@pytest.fixture(scope="module")
def lib(gpu_ctx):
    return C.CDLL(alac_amd.LIB_PATH)


def call_mix(lib, depth, p_in, p_u, p_v, n, mixres, p_sh, bs):
    vp = C.c_void_p
    if depth == 16:
        lib.mix16(vp(p_in), 2, vp(p_u), vp(p_v), n, 2, mixres)
    elif depth == 20:
        lib.mix20(vp(p_in), 2, vp(p_u), vp(p_v), n, 2, mixres)
    elif depth == 24:
        lib.mix24(vp(p_in), 2, vp(p_u), vp(p_v), n, 2, mixres, vp(p_sh), bs)
    else:
        lib.mix32(vp(p_in), 2, vp(p_u), vp(p_v), n, 2, mixres, vp(p_sh), bs)


@pytest.mark.parametrize("depth,bs", [(16, 0), (20, 0), (24, 1), (24, 0), (32, 2), (32, 1)])
@pytest.mark.parametrize("mixres", [0, 1, 3, 4])
@pytest.mark.parametrize("device", [False, True])
def test_mix_matches_oracle(lib, oracle, depth, bs, mixres, device):
    n = 4096 if device else 1000
    fmt = alac_amd.make_format(n, depth, 2, 44100)
    pcm = alac_amd.synth_pcm(3 + mixres, 1, fmt)  # class 3 / 4 / 6 / 7 signals
    if depth == 32 and bs == 1:
        pcm = pcm.copy()
        pcm.view(np.int32)[:] >>= 8  # keep the "internal width < 32" contract of mix32
    want_u, want_v, want_sh = oracle.mix(pcm, depth, n, 2, mixres, bs)
    writes_shift = (depth == 24 and bs != 0) or (depth == 32 and (mixres != 0 or bs != 0))
    if device:
        d_in = torch.from_numpy(pcm).cuda()
        d_u = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_v = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_sh = torch.zeros(2 * n, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        call_mix(lib, depth, d_in.data_ptr(), d_u.data_ptr(), d_v.data_ptr(), n, mixres, d_sh.data_ptr(), bs)
        torch.cuda.synchronize()
        u, v, sh = d_u.cpu().numpy(), d_v.cpu().numpy(), d_sh.cpu().numpy().view(np.uint16)
    else:
        u, v, sh = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(2 * n, np.uint16)
        call_mix(lib, depth, pcm.ctypes.data, u.ctypes.data, v.ctypes.data, n, mixres, sh.ctypes.data, bs)
    assert np.array_equal(u, want_u) and np.array_equal(v, want_v)
    if writes_shift:
        assert np.array_equal(sh, want_sh)
    else:
        assert not sh.any()  # untouched, as in codec/matrix_enc.cu:293-323


def test_mix_honours_the_channel_stride(lib, oracle):
    # stereo pair taken out of a 6-channel interleaved buffer (the multichannel element loop hands mixNN a stride)
    n, ch = 777, 6
    rng = np.random.default_rng(5)
    x = rng.integers(-30000, 30000, (n, ch), dtype=np.int16)
    pair = np.ascontiguousarray(x[:, 2:4])
    want_u, want_v, _ = oracle.mix(pair.view(np.uint8).ravel(), 16, n, 2, 2, 0)
    u, v = np.zeros(n, np.int32), np.zeros(n, np.int32)
    lib.mix16(C.c_void_p(x.ctypes.data + 2 * 2), ch, C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data), n, 2, 2)
    assert np.array_equal(u, want_u) and np.array_equal(v, want_v)


@pytest.mark.parametrize("stride", [1, 2])
def test_copy20_to_predictor(lib, stride):
    n = 1500
    rng = np.random.default_rng(20)
    vals = rng.integers(-(1 << 19), 1 << 19, n * stride)
    raw = np.zeros(n * stride * 3, np.uint8)
    w = (vals.astype(np.int64) << 4) & 0xFFFFFF  # 20 bits left-justified in 3 bytes
    raw[0::3], raw[1::3], raw[2::3] = w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF
    out = np.zeros(n, np.int32)
    lib.copy20ToPredictor(C.c_void_p(raw.ctypes.data), stride, C.c_void_p(out.ctypes.data), n)
    assert np.array_equal(out, vals[::stride].astype(np.int32))
