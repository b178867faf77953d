"""Many EQUAL chained segments side by side — the shape of `alacconvert --batch` (files x packets) and of
tools/chain_timing.py — in the tiny-batch regime (four lanes per chain, consecutive packet positions overlapped, final coder
split over two waves).  64 segments x 32 packets is the shape at which a development build of round 2 raised a GPU memory
access fault (the cause); 1024 x 32 = 2048 chains sits at the upper end of the regime, 1100 x 3 is
ragged against every 16- / 32- / 64-chain wave boundary.  Every packet is compared with the oracle's chain."""
import numpy as np
import pytest

import alac_amd

pytestmark = pytest.mark.gpu


def _check(gpu_ctx, oracle, fmt, nseg, per, sample_segments, expect_regime="tiny"):
    import torch
    n = nseg * per
    # with default options these shapes run in the tiny-batch regime; tests/test_gpu_variants.py runs them in the others
    assert gpu_ctx.regime(fmt, nseg) in (expect_regime, "latency", "throughput", "lane", "stagewise")
    pcm = alac_amd.synth_pcm(3, n, fmt)
    seg_first = torch.arange(0, n + 1, per, dtype=torch.int32).cuda()
    state = torch.zeros((nseg, 64), dtype=torch.int16).cuda()
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, seg_first=seg_first, state=state)
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    assert offs[-1] == len(stream)
    enc = oracle.encoder(fmt.frame_size, fmt.bit_depth, fmt.num_channels)
    for s in sample_segments:
        a, b = s * per, (s + 1) * per
        enc.reset()
        ref, rs = enc.encode_stream(pcm[a * fmt.packet_bytes:b * fmt.packet_bytes], per * fmt.frame_size, 0)
        assert np.array_equal(sizes[a:b], rs), s
        assert np.array_equal(stream[offs[a]:offs[b]], ref), s
        assert np.array_equal(state[s].cpu().numpy(), enc.get_state()), s


def test_64_segments_x_32_packets(gpu_ctx, oracle):
    _check(gpu_ctx, oracle, alac_amd.make_format(4096, 16, 2), 64, 32, range(64))


def test_1024_segments_x_32_packets(gpu_ctx, oracle):
    _check(gpu_ctx, oracle, alac_amd.make_format(4096, 16, 2), 1024, 32, [0, 1, 7, 8, 15, 16, 31, 32, 33, 511, 512, 1000, 1022, 1023])


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (16, 1)])
def test_ragged_segment_count(gpu_ctx, oracle, depth, channels):
    nseg = 1100 if channels == 2 else 2100
    _check(gpu_ctx, oracle, alac_amd.make_format(1024, depth, channels), nseg, 3,
           [0, 1, 2, 7, 8, 15, 16, 17, 31, 32, 63, 64, nseg // 2, nseg - 18, nseg - 17, nseg - 16, nseg - 2, nseg - 1])


@pytest.mark.parametrize("depth,channels,nseg", [(16, 2, 3000), (16, 2, 5632), (24, 2, 11500), (16, 1, 9000)])
def test_chained_segments_in_the_wider_four_lane_windows(gpu_ctx, oracle, depth, channels, nseg):
    """Round 4 widened the four-lanes-per-chain regime (up to 11 264 chains, and 21 761..34 816): chained segments at those
    counts now run with overlapped packet positions and WITHOUT the split final coder (its buffers stop at 4096 chains) — a
    combination the shapes above never reach.  3 packets per segment, 1024-sample frames, sampled segments against the oracle's
    chains, at a count inside the first window, at its edge, inside the second window, and a mono batch."""
    per = 3
    _check(gpu_ctx, oracle, alac_amd.make_format(1024, depth, channels), nseg, per,
           [0, 1, 2, 15, 16, 31, 32, 63, 64, 65, nseg // 3, nseg // 2, nseg - 65, nseg - 64, nseg - 17, nseg - 2, nseg - 1])


def test_segment_bound_instead_of_read_back(gpu_ctx, oracle):
    """alac_hip_encode_segmented: with the caller's bound on the segment length nothing is read back; an over-estimate gives the
    same bytes, a table that contradicts the bound fails the next synchronize (kALAC_ParamError) instead of encoding garbage"""
    import torch
    from alac_amd.capi import AlacError
    fmt = alac_amd.make_format(1024, 16, 2)
    seg_first = [0, 3, 4, 9, 12]
    n = seg_first[-1]
    pcm = alac_amd.synth_pcm(11, n, fmt)
    d = torch.from_numpy(pcm).cuda()
    sf = torch.tensor(seg_first, dtype=torch.int32).cuda()
    ref, ref_sizes = gpu_ctx.encode_to_host(fmt, d, n, seg_first=sf)
    for bound in (5, 8, 1000):
        s, z = gpu_ctx.encode_to_host(fmt, d, n, seg_first=sf, max_segment_packets=bound)
        assert np.array_equal(z, ref_sizes) and np.array_equal(s, ref), bound
    enc = oracle.encoder(1024, 16, 2)
    off = 0
    for a, b in zip(seg_first[:-1], seg_first[1:]):
        enc.reset()
        want, _ = enc.encode_stream(pcm[a * fmt.packet_bytes:b * fmt.packet_bytes], (b - a) * 1024, 0)
        assert np.array_equal(ref[off:off + len(want)], want)
        off += len(want)
    gpu_ctx.encode(fmt, d, n, seg_first=sf, max_segment_packets=4)  # the third segment has 5 packets
    with pytest.raises(AlacError) as ei:
        gpu_ctx.synchronize()
    assert ei.value.code == -50 and "max_segment_packets" in str(ei.value)
    s, z = gpu_ctx.encode_to_host(fmt, d, n, seg_first=sf, max_segment_packets=5)  # the error is consumed
    assert np.array_equal(s, ref)


@pytest.mark.parametrize("variant", ["default", "lane", "unfolded"])
@pytest.mark.parametrize("table", ["too_long", "past_the_batch", "descending", "bound_too_small_big_batch"])
def test_refused_segment_table_touches_nothing(table, variant):
    """a segment table that contradicts the caller's bound is only DISCOVERED on the device, with every later launch
    already enqueued behind the check.  Fresh context, workspace / size / offset / output buffers poisoned: the call must fail
    with kALAC_ParamError at the synchronize and must not have written one byte of packet data (no index formed from a bad
    entry, nothing packed from stale records)."""
    import torch
    from alac_amd.capi import AlacError
    ctx = alac_amd.Context(0)
    if variant == "lane":
        ctx.set_option("encoder_lane", 1)
    elif variant == "unfolded":
        ctx.set_option("fold", 0)
    fmt = alac_amd.make_format(256, 16, 2)
    if table == "too_long":
        seg_first, n, bound = [0, 3, 4, 9, 12], 12, 4
    elif table == "past_the_batch":
        seg_first, n, bound = [0, 3, 4, 0x7fffff00, 12], 12, 5
    elif table == "descending":
        seg_first, n, bound = [0, 9, 4, 3, 12], 12, 12
    else:
        n, bound = 3000, 2
        seg_first = list(range(0, n - 200, 2)) + [n]  # the last segment has 202 packets
    d = ctx.synth_pcm(3, n, fmt)
    sf = torch.tensor(seg_first, dtype=torch.int32).cuda()
    nseg = len(seg_first) - 1
    wsb = int(ctx.lib.alac_hip_encode_workspace_bytes(fmt, n, nseg))
    ctx._ws = torch.full((wsb,), 0xA5, dtype=torch.uint8, device="cuda")
    bufs = ctx.encode_buffers(fmt, n)
    bufs["out"].fill_(0x5A)
    bufs["sizes"].fill_(0x7fff0000)
    bufs["offsets"].fill_(0x7fff000000000000)
    torch.cuda.synchronize()
    ctx.encode(fmt, d, n, seg_first=sf, bufs=bufs, max_segment_packets=bound)
    with pytest.raises(AlacError) as ei:
        ctx.synchronize()
    assert ei.value.code == -50
    torch.cuda.synchronize()
    assert bool((bufs["out"] == 0x5A).all()), "packet bytes were written from an unvalidated table"
    assert int(bufs["offsets"][-1].item()) == 0
    # and the context is usable afterwards
    good = torch.arange(0, n + 1, dtype=torch.int32).cuda()
    s, z = ctx.encode_to_host(fmt, d, n, seg_first=good, max_segment_packets=1)
    s2, z2 = ctx.encode_to_host(fmt, d, n)
    assert np.array_equal(s, s2) and np.array_equal(z, z2)
    ctx.close()
