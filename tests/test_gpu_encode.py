"""GPU parity: HIP encode path vs the CPU oracle, through the C-ABI.  Bit-exact (integer/byte work)."""
import numpy as np
import pytest

import alac_amd

pytestmark = pytest.mark.gpu


def _oracle_stream(oracle, fmt, pcm, total_samples, segment_packets):
    enc = oracle.encoder(fmt.frame_size, fmt.bit_depth, fmt.num_channels, fmt.sample_rate)
    return enc.encode_stream(pcm, total_samples, segment_packets)


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (20, 2), (32, 2), (16, 1), (24, 1), (20, 1), (32, 1)])
def test_independent_packets_match_oracle(gpu_ctx, oracle, depth, channels):
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    n = 48
    pcm = alac_amd.synth_pcm(0, n, fmt)
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n)
    ref, ref_sizes = _oracle_stream(oracle, fmt, pcm, n * 4096, 1)
    assert np.array_equal(sizes, ref_sizes)
    assert np.array_equal(stream, ref)


SIZES = [1, 2, 7, 8, 9, 31, 32, 39, 40, 63, 64, 71, 72, 100, 255, 287, 288, 289, 404, 1904, 3544, 4095, 4096]


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (16, 1), (32, 1), (20, 2)])
def test_partial_packets_match_oracle(gpu_ctx, oracle, depth, channels):
    """ragged packets, incl. N/8 <= numactive and N < 8 (empty search passes)"""
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    n = len(SIZES)
    pcm = alac_amd.synth_pcm(2, n, fmt)
    ns = torch.tensor(SIZES, dtype=torch.int32).cuda()
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, num_samples=ns)
    enc = oracle.encoder(4096, depth, channels)
    off = 0
    for p, N in enumerate(SIZES):
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + N * fmt.bytes_per_frame], N)
        assert sizes[p] == len(pk), (p, N)
        assert np.array_equal(stream[off:off + len(pk)], pk), (p, N)
        off += len(pk)
    assert off == len(stream)


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (16, 1)])
def test_chained_segments_and_state(gpu_ctx, oracle, depth, channels):
    """segments of different lengths chained through the coefficient state; state out == oracle state;
    a second call continuing from that state == one long chain."""
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    seg_first = [0, 1, 4, 9, 10, 17]
    n = seg_first[-1]
    pcm = alac_amd.synth_pcm(40, n, fmt)
    d_pcm = torch.from_numpy(pcm).cuda()
    sf = torch.tensor(seg_first, dtype=torch.int32).cuda()
    state = torch.zeros((len(seg_first) - 1, 64), dtype=torch.int16).cuda()
    stream, sizes = gpu_ctx.encode_to_host(fmt, d_pcm, n, seg_first=sf, state=state)
    off = 0
    for s in range(len(seg_first) - 1):
        a, b = seg_first[s], seg_first[s + 1]
        enc = oracle.encoder(4096, depth, channels)
        ref, rs = enc.encode_stream(pcm[a * fmt.packet_bytes:b * fmt.packet_bytes], (b - a) * 4096, 0)
        assert np.array_equal(sizes[a:b], rs)
        assert np.array_equal(stream[off:off + len(ref)], ref), s
        assert np.array_equal(state[s].cpu().numpy(), enc.get_state()), s
        off += len(ref)
    # continue every segment with 2 more packets from the saved state
    more = alac_amd.synth_pcm(80, 2 * (len(seg_first) - 1), fmt)
    sf2 = torch.arange(0, 2 * (len(seg_first) - 1) + 1, 2, dtype=torch.int32).cuda()
    st2 = state.clone()
    stream2, sizes2 = gpu_ctx.encode_to_host(fmt, torch.from_numpy(more).cuda(), 2 * (len(seg_first) - 1),
                                             seg_first=sf2, state=st2, state_in=True)
    off = 0
    for s in range(len(seg_first) - 1):
        enc = oracle.encoder(4096, depth, channels)
        enc.set_state(state[s].cpu().numpy())
        ref, _ = enc.encode_stream(more[2 * s * fmt.packet_bytes:(2 * s + 2) * fmt.packet_bytes], 2 * 4096, 0)
        assert np.array_equal(stream2[off:off + len(ref)], ref), s
        off += len(ref)


@pytest.mark.parametrize("kind,ch", [("stereo", 2), ("mono", 1)])
def test_golden_packet_fixtures(gpu_ctx, kind, ch):
    """excerpts of the reference's audio files; expected packets came from the reference's own stage
    objects (tests/golden/make_golden.py)"""
    import os
    import torch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "packets.npz"))
    fmt = alac_amd.make_format(4096, 16, ch)
    pcm = z[f"{kind}_pcm"]
    total = len(pcm) // fmt.bytes_per_frame
    n = (total + 4095) // 4096
    padded = np.zeros(n * fmt.packet_bytes, np.uint8)
    padded[:len(pcm)] = pcm
    ns = torch.tensor([4096] * (n - 1) + [total - 4096 * (n - 1)], dtype=torch.int32).cuda()
    d = torch.from_numpy(padded).cuda()
    s, sz = gpu_ctx.encode_to_host(fmt, d, n, num_samples=ns)
    assert np.array_equal(sz, z[f"{kind}_indep_sizes"]) and np.array_equal(s, z[f"{kind}_indep_stream"])
    sf = torch.tensor([0, n], dtype=torch.int32).cuda()
    s, sz = gpu_ctx.encode_to_host(fmt, d, n, num_samples=ns, seg_first=sf)
    assert np.array_equal(sz, z[f"{kind}_chained_sizes"]) and np.array_equal(s, z[f"{kind}_chained_stream"])


def test_full_size_properties_10k(gpu_ctx, oracle):
    """BASELINE configs[1] at full size: size-independent properties (decode round trip, idempotence,
    scan consistency) + oracle bytes on a sampled subset."""
    import torch
    fmt = alac_amd.make_format(4096, 16, 2)
    n = 10000
    pcm = alac_amd.synth_pcm(0, n, fmt)
    d_pcm = torch.from_numpy(pcm).cuda()
    b = gpu_ctx.encode(fmt, d_pcm, n)
    gpu_ctx.synchronize()
    offs = b["offsets"].cpu().numpy()
    sizes = b["sizes"].cpu().numpy().astype(np.int64)
    assert offs[0] == 0 and np.array_equal(np.diff(offs), sizes)
    total = int(offs[-1])
    stream = b["out"][:total].clone()
    b2 = gpu_ctx.encode(fmt, d_pcm, n)
    gpu_ctx.synchronize()
    assert torch.equal(b2["out"][:total], stream) and torch.equal(b2["sizes"], b["sizes"])
    out, ns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), stream, b["offsets"], n)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and bool((ns == 4096).all())
    assert torch.equal(out, d_pcm)
    host = stream.cpu().numpy()
    enc = oracle.encoder(4096, 16, 2)
    for p in list(range(0, n, 101)) + [n - 1]:
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes], 4096)
        assert np.array_equal(host[offs[p]:offs[p + 1]], pk), p


def test_rejects_bad_arguments(gpu_ctx):
    import torch
    fmt = alac_amd.make_format(4096, 12, 2)
    with pytest.raises(Exception):
        gpu_ctx.encode(fmt, torch.zeros(1 << 16, dtype=torch.uint8).cuda(), 1)


def test_throughput_regime_with_chained_segments(gpu_ctx, oracle):
    """more than 65 536 chains (the throughput regime: separate launches, class compaction, 8 taps per lane) AND several
    packets per segment: every position of every segment goes through the class layout; a sample of whole segments
    (incl. the first, the last and the ones around a 1024-segment compaction block edge) must equal the oracle's chains"""
    import torch
    frame = 256
    fmt = alac_amd.make_format(frame, 16, 2)
    nseg = 34000
    lens = np.where(np.arange(nseg) % 3 == 0, 3, 2).astype(np.int32)      # 2 or 3 packets per segment
    seg_first = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    n = int(seg_first[-1])
    pcm = alac_amd.synth_pcm(0, n, fmt)
    ns = np.full(n, frame, np.int32)
    ns[seg_first[1:] - 1] = np.where(np.arange(nseg) % 5 == 0, 100, frame)  # some segments end in a partial packet
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, num_samples=torch.from_numpy(ns).cuda(),
                                           seg_first=torch.from_numpy(seg_first).cuda())
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    enc = oracle.encoder(frame, 16, 2)
    for s in [0, 1, 2, 5, 1022, 1023, 1024, 1025, 17000, 17001, 33995, 33998, 33999]:
        enc.reset()
        for p in range(seg_first[s], seg_first[s + 1]):
            pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + ns[p] * fmt.bytes_per_frame], int(ns[p]))
            assert sizes[p] == len(pk), (s, p)
            assert np.array_equal(stream[offs[p]:offs[p + 1]], pk), (s, p)


@pytest.mark.parametrize("depth", [16, 20, 24, 32])
@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("frame", [4096, 333])
def test_packer_spans_and_seams_every_depth(gpu_ctx, oracle, depth, channels, frame):
    """The packer's fast paths (k_pack): escape payloads read with wide loads and picked by i mod 3 (24-bit) / i mod 5
    (20-bit), shift-off bytes, the words either side of every span, ragged ends — packets of full-scale noise (escape)
    and of a quiet tone (compressed), every length class, at every byte alignment the running offset produces."""
    import torch
    fmt = alac_amd.make_format(frame, depth, channels)
    lengths = [frame, frame - 1, frame - 2, frame - 3, max(frame // 3, 1), 37, 6, 5, 4, 3, 2, 1]
    ns = np.array([n for n in lengths for _ in range(2)], np.int32)  # each length once as noise, once as tone
    n = len(ns)
    rng = np.random.default_rng(depth * 100 + channels * 10 + (frame & 7))
    bpf = fmt.bytes_per_frame
    pcm = np.zeros(n * fmt.packet_bytes, np.uint8)
    top = 1 << (depth - 1)
    for p in range(n):
        N = int(ns[p])
        if p % 2 == 0:
            a = rng.integers(-top, top, (N, channels), dtype=np.int64)  # incompressible: the escape path
        else:
            t = np.arange(N)[:, None]
            a = np.round(top / 64 * np.sin(2 * np.pi * 440.0 * t / 44100.0 + np.arange(channels))).astype(np.int64)
            a += rng.integers(-2, 3, a.shape)
        if depth == 16:
            b = a.astype("<i2").view(np.uint8).reshape(-1)
        elif depth == 32:
            b = a.astype("<i4").view(np.uint8).reshape(-1)
        else:
            if depth == 20:
                a = a << 4
            b = (a & 0xffffff).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3].reshape(-1).copy()
        pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + N * bpf] = b
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, num_samples=torch.from_numpy(ns).cuda())
    enc = oracle.encoder(frame, depth, channels)
    off, escapes = 0, 0
    for p in range(n):
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + int(ns[p]) * bpf], int(ns[p]))
        escapes += enc.last_info()["escape"]
        assert sizes[p] == len(pk), (p, int(ns[p]))
        assert np.array_equal(stream[off:off + len(pk)], pk), (p, int(ns[p]))
        off += len(pk)
    assert off == len(stream)
    assert escapes >= 4  # the noise packets of useful length really took the uncompressed path


@pytest.mark.parametrize("opts", [{}, {"thru": 1}, {"narrow": 0}, {"fused": 0}],
                         ids=["default", "throughput", "two-lane", "stagewise"])
def test_tiny_frames_ending_in_zero_runs(gpu_ctx, oracle, opts):
    """Frames so short that the numUV count has no stale tail (N / 8 <= max(N / 32, numUV + 1), i.e. N < 80) and whose residuals
    end in zeros, so that every bit count ends with an OPEN zero run that the end of the stream has to close (ag_enc.c:351).
    Round 3's in-lane count (k_search2_lane) only closed it for lanes that had a tail: 32 of 3000 fuzz seeds, all with
    17-sample stereo frames, chose another numUV than the reference."""
    import torch
    rng = np.random.default_rng(7)
    for frame in (8, 9, 16, 17, 24, 31, 33, 40, 64, 72, 79, 80, 81, 96):
        fmt = alac_amd.make_format(frame, 16, 2)
        n = 96
        x = (rng.standard_normal((n, frame, 2)) * rng.choice([0, 3, 300, 9000], (n, 1, 1))).round().astype(np.int64)
        x[:, frame // 3:, :] //= 64          # quiet ...
        x[::2, frame // 2:, :] = 0           # ... or silent tails: the residuals end in zero runs
        x[1::4, :, 1] = x[1::4, :, 0]        # identical channels in some packets (v = 0 under mixing)
        pcm = np.clip(x, -32768, 32767).astype("<i2").view(np.uint8).reshape(-1).copy()
        with gpu_ctx.options(**opts):
            stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n)
        ref, ref_sizes = _oracle_stream(oracle, fmt, pcm, n * frame, 1)
        assert np.array_equal(sizes, ref_sizes), (frame, np.nonzero(sizes != ref_sizes)[0][:8])
        assert np.array_equal(stream, ref), frame


@pytest.mark.parametrize("depth,channels,n,regime", [(16, 2, 5632, "tiny"), (16, 2, 5700, "latency"), (16, 2, 10880, "latency"),
                                                     (16, 2, 11000, "tiny"), (24, 2, 12000, "tiny"), (16, 2, 17408, "tiny"),
                                                     (16, 2, 17500, "latency"), (16, 1, 10240, "tiny"), (16, 1, 10300, "latency"),
                                                     (16, 1, 21800, "tiny"), (24, 1, 26112, "tiny"), (16, 1, 26200, "latency")])
def test_regime_windows_below_the_throughput_regime(gpu_ctx, oracle, depth, channels, n, regime):
    """The automatic choice between four and two lanes per chain (v1_narrow_regime): four lanes up to 11 264 chains, two while
    their workers have a SIMD each (21 760 chains), four again up to 34 816 chains, two from there to the throughput regime.
    At both edges of every window: the regime the library reports, packets equal to the oracle's on a sample that covers all
    eight signal classes, and the whole batch decodes back to the input."""
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    if gpu_ctx.get_option("narrow") == -1 and gpu_ctx.get_option("thru") == -1 and gpu_ctx.get_option("fused") != 0 \
            and not gpu_ctx.get_option("encoder_lane"):
        assert gpu_ctx.regime(fmt, n) == regime
    d_pcm = gpu_ctx.synth_pcm(7, n, fmt)
    b = gpu_ctx.encode(fmt, d_pcm, n)
    gpu_ctx.synchronize()
    offs = b["offsets"].cpu().numpy()
    idx = sorted(set(list(range(0, n, 397)) + [0, 1, 63, 64, n - 65, n - 64, n - 1]))
    assert {(7 + p) % 8 for p in idx} == set(range(8))
    enc = oracle.encoder(4096, depth, channels)
    for p in idx:
        enc.reset()
        want = enc.encode_packet(alac_amd.synth_pcm(7 + p, 1, fmt), 4096)
        assert np.array_equal(b["out"][int(offs[p]):int(offs[p + 1])].cpu().numpy(), want), p
    out, ns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), b["out"], b["offsets"], n, zero_fill=False)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and torch.equal(out, d_pcm)
