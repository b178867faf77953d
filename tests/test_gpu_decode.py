"""GPU parity, decode direction: HIP decode of oracle-encoded streams == the PCM that was encoded and ==
the oracle decoder, incl. partial packets, escapes, shift-off bytes, every bit depth."""
import numpy as np
import pytest

import alac_amd

pytestmark = pytest.mark.gpu

SIZES = [4096, 1, 7, 8, 9, 40, 72, 100, 288, 404, 1904, 3544, 4095, 4096, 4096, 33]


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (20, 2), (32, 2), (16, 1), (24, 1), (20, 1), (32, 1)])
def test_decode_oracle_streams(gpu_ctx, oracle, depth, channels):
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    n = len(SIZES)
    pcm = alac_amd.synth_pcm(0, n, fmt)
    enc = oracle.encoder(4096, depth, channels)
    pkts = []
    for p, N in enumerate(SIZES):
        if p % 3 == 0:
            enc.reset()  # mix chained and fresh coefficient rows: the decoder must not care
        pkts.append(enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + N * fmt.bytes_per_frame], N))
    stream = np.concatenate(pkts)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in pkts])]).astype(np.int64)
    out, ns, st, f2 = gpu_ctx.decode(oracle.encoder(4096, depth, channels).cookie(), torch.from_numpy(stream).cuda(),
                                     torch.from_numpy(offs).cuda(), n)
    gpu_ctx.synchronize()
    assert (f2.frame_size, f2.bit_depth, f2.num_channels) == (4096, depth, channels)
    assert st.cpu().tolist() == [0] * n and ns.cpu().tolist() == SIZES
    out = out.cpu().numpy()
    for p, N in enumerate(SIZES):
        a = p * fmt.packet_bytes
        assert np.array_equal(out[a:a + N * fmt.bytes_per_frame], pcm[a:a + N * fmt.bytes_per_frame]), (p, N)


def test_decode_flags_corrupt_packets(gpu_ctx, oracle):
    import torch
    fmt = alac_amd.make_format(4096, 16, 2)
    pcm = alac_amd.synth_pcm(3, 2, fmt)
    enc = oracle.encoder(4096, 16, 2)
    good = enc.encode_packet(pcm[:fmt.packet_bytes], 4096)
    bad = good.copy()
    bad[1] |= 0x10  # non-zero "unused header" bits -> kALAC_ParamError (ALACDecoder.cu:768)
    trunc = good[:len(good) // 2]
    stream = np.concatenate([good, bad, trunc])
    offs = np.array([0, len(good), 2 * len(good), 2 * len(good) + len(trunc)], np.int64)
    out, ns, st, _ = gpu_ctx.decode(enc.cookie(), torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), 3)
    gpu_ctx.synchronize()
    st = st.cpu().tolist()
    assert st[0] == 0 and st[1] == -50 and st[2] == -50
    assert np.array_equal(out[:fmt.packet_bytes].cpu().numpy(), pcm[:fmt.packet_bytes])


@pytest.mark.parametrize("frame", [64, 2048])
def test_more_packets_than_a_grid_dimension(gpu_ctx, oracle, frame):
    """70 000 packets in one call (a launch grid's y dimension stops at 65 535): encode -> decode round trip on
    the GPU, oracle bytes on a sample of the packets"""
    import torch
    fmt = alac_amd.make_format(frame, 16, 2)
    n = 70000
    pcm = alac_amd.synth_pcm(0, n, fmt)
    d_pcm = torch.from_numpy(pcm).cuda()
    b = gpu_ctx.encode(fmt, d_pcm, n)
    gpu_ctx.synchronize()
    offs = b["offsets"].cpu().numpy()
    stream = b["out"][:int(offs[-1])].cpu().numpy()
    enc = oracle.encoder(frame, 16, 2)
    for p in (0, 1, 7, 65534, 65535, 65536, 69999):
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes], frame)
        assert np.array_equal(stream[offs[p]:offs[p + 1]], pk), p
    out, ns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), b["out"], b["offsets"], n)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and int((ns != frame).sum()) == 0
    assert torch.equal(out, d_pcm)
