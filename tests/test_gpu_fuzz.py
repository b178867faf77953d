"""Randomised parity: frame sizes (odd ones too, which defeat the vectorised staging), depths, channels, ragged
packets and segment layouts drawn from a seeded generator; every packet of every case must equal the oracle's, and
decode back to the input.  Small on purpose (seconds), wide on shapes."""
import numpy as np
import pytest

import alac_amd

pytestmark = pytest.mark.gpu


def noisy_music(rng, frames, channels, depth):
    t = np.arange(frames)
    amp = float(1 << (depth - 3))
    out = []
    for c in range(channels):
        f0 = rng.uniform(50, 3000)
        x = amp * 0.5 * np.sin(2 * np.pi * f0 * t / 44100.0 + rng.uniform(0, 6))
        x += rng.standard_normal(frames) * amp * rng.choice([0.0, 0.001, 0.05, 0.5])
        if rng.random() < 0.2:
            x[: frames // 2] = 0        # long zero runs
        out.append(x)
    a = np.clip(np.round(np.stack(out, axis=1)), -(1 << (depth - 1)), (1 << (depth - 1)) - 1).astype(np.int64)
    if depth == 16:
        return a.astype("<i2").view(np.uint8).reshape(-1)
    if depth == 32:
        return a.astype("<i4").view(np.uint8).reshape(-1)
    if depth == 20:
        a = a << 4                       # 20-bit samples sit in the top of 3 bytes
    return (a & 0xffffff).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3].reshape(-1).copy()


@pytest.mark.parametrize("seed", range(48))
def test_random_layouts_match_oracle_and_round_trip(gpu_ctx, oracle, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    depth = int(rng.choice([16, 16, 24, 20, 32]))
    channels = int(rng.choice([1, 2, 2]))
    frame = int(rng.choice([4096, 4096, 1024, 512, 100, 333, 4095, 64, 17, 2048]))
    fmt = alac_amd.make_format(frame, depth, channels)
    nseg = int(rng.integers(1, 7))
    seg_len = rng.integers(1, 6, nseg)
    seg_first = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.int32)
    n = int(seg_first[-1])
    ns = np.full(n, frame, np.int32)
    for s in range(nseg):                 # the last packet of a segment may be partial, as in a file
        if rng.random() < 0.6:
            ns[seg_first[s + 1] - 1] = int(rng.integers(1, frame + 1))
    bpf = fmt.bytes_per_frame
    pcm = np.zeros(n * fmt.packet_bytes, np.uint8)
    for p in range(n):
        pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + ns[p] * bpf] = noisy_music(rng, int(ns[p]), channels, depth)
    d_pcm = torch.from_numpy(pcm).cuda()
    stream, sizes = gpu_ctx.encode_to_host(fmt, d_pcm, n, num_samples=torch.from_numpy(ns).cuda(),
                                           seg_first=torch.from_numpy(seg_first).cuda())
    enc = oracle.encoder(frame, depth, channels)
    off = 0
    for s in range(nseg):
        enc.reset()
        for p in range(seg_first[s], seg_first[s + 1]):
            pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + ns[p] * bpf], int(ns[p]))
            assert sizes[p] == len(pk), (seed, depth, channels, frame, p, int(ns[p]))
            assert np.array_equal(stream[off:off + len(pk)], pk), (seed, depth, channels, frame, p, int(ns[p]))
            off += len(pk)
    assert off == len(stream)
    # decode direction on the same stream
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])).cuda()
    out, dns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), torch.from_numpy(stream).cuda(), offs, n)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and np.array_equal(dns.cpu().numpy(), ns)
    got = out.cpu().numpy()
    for p in range(n):
        a, b = p * fmt.packet_bytes, p * fmt.packet_bytes + ns[p] * bpf
        assert np.array_equal(got[a:b], pcm[a:b]), (seed, p)


@pytest.mark.parametrize("frame,depth,channels", [(16384, 16, 2), (65536, 24, 2), (8192, 16, 1)])
def test_large_frames(gpu_ctx, oracle, frame, depth, channels):
    """frame sizes well beyond the default 4096 (many tiles, long planes)"""
    import torch
    rng = np.random.default_rng(frame)
    fmt = alac_amd.make_format(frame, depth, channels)
    n = 3
    pcm = np.concatenate([noisy_music(rng, frame, channels, depth) for _ in range(n)])
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n)
    enc = oracle.encoder(frame, depth, channels)
    ref, ref_sizes = enc.encode_stream(pcm, n * frame, segment_packets=1)
    assert np.array_equal(sizes, ref_sizes) and np.array_equal(stream, ref)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])).cuda()
    out, dns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), torch.from_numpy(stream).cuda(), offs, n)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and np.array_equal(out.cpu().numpy(), pcm)


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (16, 1)])
def test_many_uneven_segments(gpu_ctx, oracle, depth, channels):
    """several waves of chained segments of unequal length: at the later packet positions most lanes are idle,
    at the first every lane works (the idle-lane paths of the predictor and coder kernels)"""
    import torch
    rng = np.random.default_rng(77 + depth + channels)
    frame = 1024
    fmt = alac_amd.make_format(frame, depth, channels)
    nseg = 150
    seg_len = rng.integers(1, 6, nseg)
    seg_first = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.int32)
    n = int(seg_first[-1])
    ns = np.full(n, frame, np.int32)
    for s in range(nseg):
        if rng.random() < 0.3:
            ns[seg_first[s + 1] - 1] = int(rng.integers(1, frame + 1))
    bpf = fmt.bytes_per_frame
    pcm = np.zeros(n * fmt.packet_bytes, np.uint8)
    for p in range(n):
        pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + ns[p] * bpf] = noisy_music(rng, int(ns[p]), channels, depth)
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, num_samples=torch.from_numpy(ns).cuda(),
                                           seg_first=torch.from_numpy(seg_first).cuda())
    enc = oracle.encoder(frame, depth, channels)
    off = 0
    for s in range(nseg):
        enc.reset()
        for p in range(seg_first[s], seg_first[s + 1]):
            pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + ns[p] * bpf], int(ns[p]))
            assert sizes[p] == len(pk) and np.array_equal(stream[off:off + len(pk)], pk), (s, p)
            off += len(pk)
    assert off == len(stream)
