"""A lost in-launch producer -> consumer hand-off must be an ERROR, not corrupt output:
the context option "debug_lose_handoff" (default: ALAC_HIP_DEBUG_LOSE_HANDOFF=1) makes the producers of the fused launches never publish; the consumers' bounded waits
run out, the context's error word is raised, the next synchronize returns kALAC_MemFullError (-108) and the decoder
marks the packets kALAC_ParamError (-50).  Without the switch the same calls are bit-exact."""
import numpy as np
import pytest
import torch

import alac_amd
from alac_amd.capi import AlacError

pytestmark = pytest.mark.gpu


def test_encode_reports_a_lost_handoff(gpu_ctx, oracle):
    fmt = alac_amd.make_format(4096, 16, 2)
    n = 192
    pcm = alac_amd.synth_pcm(0, n, fmt)
    d_pcm = torch.from_numpy(pcm).cuda()
    with gpu_ctx.options(debug_lose_handoff=1):
        gpu_ctx.encode(fmt, d_pcm, n)
        with pytest.raises(AlacError) as ei:
            gpu_ctx.synchronize()
        assert ei.value.code == -108 and "hand-off" in str(ei.value)
        # the host-buffer entry point surfaces it as its own return value
        lib = gpu_ctx.lib
        import ctypes as C
        out = np.zeros(int(lib.alac_hip_encode_max_output_bytes(C.byref(fmt), n)), np.uint8)
        sizes = np.zeros(n, np.uint32)
        total = C.c_uint64(0)
        rc = lib.alac_hip_encode_host(gpu_ctx.h, C.byref(fmt), pcm.ctypes.data, n * 4096, 1, None, 0, out.ctypes.data, out.size,
                                      sizes.ctypes.data, C.byref(total))
        assert rc == -108
    # the error is consumed: the context works again once the producers publish
    stream, sizes = gpu_ctx.encode_to_host(fmt, d_pcm, n)
    ref, ref_sizes = oracle.encoder(4096, 16, 2).encode_stream(pcm, n * 4096, segment_packets=1)
    assert np.array_equal(sizes, ref_sizes) and np.array_equal(stream, ref)


def test_decode_reports_a_lost_handoff(gpu_ctx, oracle):
    fmt = alac_amd.make_format(4096, 16, 2)
    n = 96
    pcm = alac_amd.synth_pcm(0, n, fmt)
    enc = oracle.encoder(4096, 16, 2)
    stream, sizes = enc.encode_stream(pcm, n * 4096, segment_packets=1)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])).cuda()
    d_stream = torch.from_numpy(stream).cuda()
    with gpu_ctx.options(debug_lose_handoff=1):
        out, ns, st, _ = gpu_ctx.decode(enc.cookie(), d_stream, offs, n)
        with pytest.raises(AlacError) as ei:
            gpu_ctx.synchronize()
        assert ei.value.code == -108
        st = st.cpu().numpy()
        coded = np.array([p % 8 != 1 for p in range(n)])  # class 1 = escape packets: no entropy -> predictor hand-off
        assert (st[coded] == -50).all()
    out, ns, st, _ = gpu_ctx.decode(enc.cookie(), d_stream, offs, n)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and np.array_equal(out.cpu().numpy(), pcm)
