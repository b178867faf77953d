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


@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("n", [1, 47, 48, 49, 95, 96, 97, 145])
def test_packet_counts_around_the_entropy_wave_size(gpu_ctx, channels, n):
    """the fused decode launch gives an entropy wave 48 packets and its three follower waves 32 chains each: batch sizes on
    both sides of those boundaries, mono (48 chains per workgroup: one and a half followers) and stereo, round trip exact"""
    import torch
    fmt = alac_amd.make_format(512, 16, channels)
    d_pcm = gpu_ctx.synth_pcm(3, n, fmt)
    b = gpu_ctx.encode(fmt, d_pcm, n)
    out, ns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), b["out"], b["offsets"], n)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and int((ns != 512).sum()) == 0
    assert torch.equal(out, d_pcm)


@pytest.mark.parametrize("depth", [16, 24, 20, 32])
def test_large_ragged_batch_in_the_separate_launch_regime(gpu_ctx, oracle, depth):
    """40 000 stereo packets (80 000 chains: the separate-launch decode regime with its work lists) whose lengths are
    drawn per packet — full frames, lengths that are not multiples of four or eight, and lengths below 16 samples, which the
    one-lane predictor does not take (generic predictor + k_dec_unmix's leftover list) — mixed with the generator's
    silent, noisy (uncompressed) and tonal classes: every packet's valid frames must come back, oracle bytes on a sample.
    16, 20 and 24 bits take the pair form (the predictor lanes of a packet un-mix, re-attach the shifted-off bytes and
    write the PCM themselves); 32 bits (two shifted-off bytes) the one-lane predictor + k_dec_unmix."""
    import torch
    frame, n = 256, 40000
    fmt = alac_amd.make_format(frame, depth, 2)
    bpf = fmt.bytes_per_frame
    rng = np.random.default_rng(77)
    ns = np.full(n, frame, np.int32)
    pick = rng.random(n)
    ns[pick < 0.30] = rng.integers(16, frame + 1, int((pick < 0.30).sum()))
    ns[pick < 0.05] = rng.integers(1, 16, int((pick < 0.05).sum()))
    d_pcm = gpu_ctx.synth_pcm(0, n, fmt)
    d_ns = torch.from_numpy(ns).cuda()
    b = gpu_ctx.encode(fmt, d_pcm, n, num_samples=d_ns)
    gpu_ctx.synchronize()
    offs = b["offsets"].cpu().numpy()
    stream = b["out"][:int(offs[-1])].cpu().numpy()
    pcm = d_pcm.cpu().numpy()
    enc = oracle.encoder(frame, depth, 2)
    for p in list(range(0, n, 997)) + [n - 1] + [int(i) for i in np.nonzero(ns < 16)[0][:8]]:
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + ns[p] * bpf], int(ns[p]))
        assert np.array_equal(stream[offs[p]:offs[p + 1]], pk), p
    sentinel = torch.full((n * fmt.packet_bytes,), 0xA7, dtype=torch.uint8, device="cuda")
    out_bufs = (sentinel, torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"))
    out, dns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), b["out"], b["offsets"], n, out=out_bufs)
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and torch.equal(dns.to(torch.int32), d_ns)
    got = out.reshape(n, frame, bpf)
    want = d_pcm.reshape(n, frame, bpf)
    valid = (torch.arange(frame, device="cuda")[None, :] < d_ns[:, None])[:, :, None]
    bad = ((got != want) & valid).any(dim=2).any(dim=1)
    assert not bool(bad.any()), ("packets differ", torch.nonzero(bad)[:8].flatten().tolist(), ns[torch.nonzero(bad)[:8].flatten().cpu().numpy()])
    # nothing is written behind a short packet's frames
    spill = ((got != 0xA7) & ~valid).any(dim=2).any(dim=1)
    assert not bool(spill.any()), ("bytes written past the packet's frames", torch.nonzero(spill)[:8].flatten().tolist())


def _bits(value, width):
    return [(value >> (width - 1 - i)) & 1 for i in range(width)]


def _fil(count):
    """ID_FIL element with `count` payload bytes (codec/ALACDecoder.cu:1012-1027)"""
    b = _bits(6, 3)
    if count < 15:
        b += _bits(count, 4)
    else:
        b += _bits(15, 4) + _bits(count - 15 + 1, 8)
    return b + [1, 0] * (4 * count)


def _dse(count, align):
    """ID_DSE element (codec/ALACDecoder.cu:1033-1059); the caller pads to a byte boundary itself when align is set"""
    b = _bits(4, 3) + _bits(2, 4) + [align]
    b += _bits(count, 8) if count < 255 else _bits(255, 8) + _bits(count - 255, 8)
    return b


def _element_bits(pk):
    b = np.unpackbits(np.ascontiguousarray(pk, np.uint8))
    last = int(np.nonzero(b)[0][-1])
    return list(b[:last - 2])  # without ID_END and the padding


@pytest.mark.parametrize("channels", [1, 2])
def test_fill_and_data_elements_are_skipped_and_may_exceed_the_regular_stream_bound(gpu_ctx, oracle, channels):
    """a legal stream padded with ID_FIL / ID_DSE elements is longer than num_packets regular packets; the
    workspace is sized from the real stream and every packet decodes (oracle decoder = restated ALACDecoder::Decode)"""
    import torch
    depth, n = 16, 6
    fmt = alac_amd.make_format(4096, depth, channels)
    pcm = alac_amd.synth_pcm(1, n, fmt)  # packet 0 is full-scale noise: an escape packet, the largest there is
    enc = oracle.encoder(4096, depth, channels)
    cookie = enc.cookie()
    dec = oracle.decoder(cookie)
    pks = []
    for p in range(n):
        enc.reset()
        e = _element_bits(enc.encode_packet(pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes], 4096))
        if p % 3 == 0:
            b = _fil(269) + e
        elif p % 3 == 1:
            head = _fil(3) + _dse(300, 1)
            b = head + [0] * (-len(head) % 8) + [1, 1, 0, 0] * (2 * 300) + e
        else:
            b = _dse(7, 0) + [0, 1] * (4 * 7) + e
        pks.append(np.packbits(np.array(b + [1, 1, 1], np.uint8)))
    regular = alac_amd.load_library().alac_hip_encode_max_output_bytes(fmt, 1)
    assert max(len(q) for q in pks) > regular  # the case the fixed bound did not cover
    stream = np.concatenate(pks)
    offs = np.concatenate([[0], np.cumsum([len(q) for q in pks])]).astype(np.int64)
    out, ns, st, _ = gpu_ctx.decode(cookie, torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), n)
    gpu_ctx.synchronize()
    out = out.cpu().numpy()
    for p in range(n):
        want_st, want, want_n = dec.decode_packet(pks[p], fmt.bytes_per_frame)
        assert want_st == 0 and want_n == 4096
        assert int(st[p]) == 0 and int(ns[p]) == 4096, p
        a = p * fmt.packet_bytes
        assert np.array_equal(out[a:a + fmt.packet_bytes], want), p
        assert np.array_equal(want, pcm[a:a + fmt.packet_bytes]), p


def test_stream_beyond_the_workspace_fails_cleanly(gpu_ctx, oracle):
    """the decoder handed a workspace sized for regular packets only: the padded tail packets get -50, nothing is read
    out of bounds, the packets that fit still decode"""
    import ctypes as C
    import torch
    fmt = alac_amd.make_format(256, 16, 2)
    n = 8
    pcm = alac_amd.synth_pcm(1, n, fmt)
    enc = oracle.encoder(256, 16, 2)
    pks = []
    for p in range(n):
        enc.reset()
        e = _element_bits(enc.encode_packet(pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes], 256))
        pks.append(np.packbits(np.array(_fil(269) + _fil(269) + _fil(269) + _fil(269) + e + [1, 1, 1], np.uint8)))
    stream = torch.from_numpy(np.concatenate(pks)).cuda()
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum([len(q) for q in pks])]).astype(np.int64)).cuda()
    lib = gpu_ctx.lib
    ck = np.ascontiguousarray(enc.cookie(), np.uint8)
    wsb = int(lib.alac_hip_decode_workspace_bytes(C.byref(fmt), n))  # NOT the stream-sized variant
    assert int(lib.alac_hip_decode_workspace_bytes_stream(C.byref(fmt), n, int(stream.numel()))) > wsb
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    out = torch.zeros(n * fmt.packet_bytes, dtype=torch.uint8, device="cuda")
    ns = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = torch.zeros(n, dtype=torch.int32, device="cuda")
    rc = lib.alac_hip_decode(gpu_ctx.h, ck.ctypes.data, ck.size, stream.data_ptr(), offs.data_ptr(), n, ws.data_ptr(), wsb,
                             out.data_ptr(), ns.data_ptr(), st.data_ptr())
    assert rc == 0
    gpu_ctx.synchronize()
    st = st.cpu().tolist()
    if gpu_ctx.get_option("decoder_lane") or (gpu_ctx.get_option("dec_fused") == 0 and gpu_ctx.get_option("dec_direct") == 2):
        # the first-generation decoder, and the separate launches of a 16-bit stream (dec_direct), read the packets where they
        # lie (no staged copy): everything decodes
        assert st == [0] * n
        assert np.array_equal(out.cpu().numpy(), pcm)
        return
    assert st[0] == 0 and st[-1] == -50 and all(s in (0, -50) for s in st)
    assert st == sorted(st, reverse=True)  # a prefix decodes, the tail fails
    k = st.index(-50)
    assert np.array_equal(out[:k * fmt.packet_bytes].cpu().numpy(), pcm[:k * fmt.packet_bytes])


@pytest.mark.parametrize("channels", [1, 2])
def test_truncated_escape_packet(gpu_ctx, oracle, channels):
    """an uncompressed element whose fixed-width payload runs past the packet must not decode the next packet's bytes"""
    import torch
    fmt = alac_amd.make_format(4096, 16, channels)
    pcm = alac_amd.synth_pcm(1, 3, fmt)  # frame 1 = full-scale noise -> escape
    enc = oracle.encoder(4096, 16, channels)
    esc = enc.encode_packet(pcm[:fmt.packet_bytes], 4096)
    assert enc.last_info()["escape"] == 1
    enc.reset()
    ok = enc.encode_packet(pcm[2 * fmt.packet_bytes:3 * fmt.packet_bytes], 4096)
    cut = esc[:len(esc) - 100]
    pks = [cut, ok, esc]
    stream = np.concatenate(pks)
    offs = np.concatenate([[0], np.cumsum([len(q) for q in pks])]).astype(np.int64)
    out, ns, st, _ = gpu_ctx.decode(enc.cookie(), torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), 3)
    gpu_ctx.synchronize()
    assert st.cpu().tolist() == [-50, 0, 0]
    out = out.cpu().numpy()
    assert np.array_equal(out[fmt.packet_bytes:2 * fmt.packet_bytes], pcm[2 * fmt.packet_bytes:3 * fmt.packet_bytes])
    assert np.array_equal(out[2 * fmt.packet_bytes:], pcm[:fmt.packet_bytes])


def test_two_channel_stream_of_two_mono_elements(gpu_ctx, oracle):
    """a 2-channel cookie whose packets are SCE + SCE (or LFE) instead of one CPE is legal
    (codec/ALACDecoder.cu:622-756); the fast pipeline reports -4 and the gated lane decoder takes the batch"""
    import torch
    from oracle_lib import splice_elements, interleave_channels
    depth, n = 16, 5
    f1, f2 = alac_amd.make_format(4096, depth, 1), alac_amd.make_format(4096, depth, 2)
    a, b = alac_amd.synth_pcm(3, n, f1), alac_amd.synth_pcm(13, n, f1)
    stereo = alac_amd.synth_pcm(4, n, f2)

    def packets(pcm, ch):
        enc = oracle.encoder(4096, depth, ch)
        s, z = enc.encode_stream(pcm, n * 4096, 1)
        o = np.concatenate([[0], np.cumsum(z)]).astype(np.int64)
        return [s[o[p]:o[p + 1]] for p in range(n)]

    pa, pb, ps = packets(a, 1), packets(b, 1), packets(stereo, 2)
    cookie = oracle.encoder(4096, depth, 2).cookie()
    dec = oracle.decoder(cookie)
    pks = [splice_elements([(pa[p], 0), (pb[p], 1)]) if p != 2 else ps[p] for p in range(n)]  # one ordinary CPE packet among them
    stream = np.concatenate(pks)
    offs = np.concatenate([[0], np.cumsum([len(q) for q in pks])]).astype(np.int64)
    out, ns, st, _ = gpu_ctx.decode(cookie, torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), n)
    gpu_ctx.synchronize()
    out = out.cpu().numpy()
    both = interleave_channels([(a, 1), (b, 1)], depth)
    for p in range(n):
        want_st, want, want_n = dec.decode_packet(pks[p], f2.bytes_per_frame)
        assert want_st == 0 and want_n == 4096
        assert int(st[p]) == 0 and int(ns[p]) == 4096, p
        s = slice(p * f2.packet_bytes, (p + 1) * f2.packet_bytes)
        assert np.array_equal(out[s], want), p
        assert np.array_equal(want, stereo[s] if p == 2 else both[s]), p


def test_cookie_with_unusable_ag_parameters_is_refused(gpu_ctx, oracle):
    import torch
    fmt = alac_amd.make_format(4096, 16, 2)
    enc = oracle.encoder(4096, 16, 2)
    pk = enc.encode_packet(alac_amd.synth_pcm(3, 1, fmt), 4096)
    offs = torch.tensor([0, len(pk)], dtype=torch.int64, device="cuda")
    for field, value in ((8, 0), (8, 17), (6, 0)):  # kb = 0, kb = 17, pb = 0
        ck = enc.cookie().copy()
        ck[field] = value
        with pytest.raises(RuntimeError):
            gpu_ctx.decode(ck, torch.from_numpy(pk).cuda(), offs, 1)


@pytest.mark.parametrize("channels", [2, 6])
def test_decode_through_wrapped_cookies(gpu_ctx, oracle, channels):
    """the cookie as it appears outside CAF — the legacy 'frma' + 'alac' atom wrapping and the cookie taken
    out of an MP4 'stsd' sample description (ALACMagicCookieDescription.txt:177-238; ALACDecoder::Init skips the atoms,
    codec/ALACDecoder.cu:123-134) — must decode on the GPU exactly like the bare cookie."""
    import torch
    from container_lib import Container
    cont = Container()
    n = 6
    fmt = alac_amd.make_format(4096, 16, channels)
    if channels == 2:
        pcm = alac_amd.synth_pcm(2, n, fmt)
    else:
        from oracle_lib import interleave_channels
        f1, f2 = alac_amd.make_format(4096, 16, 1), alac_amd.make_format(4096, 16, 2)
        pcm = interleave_channels([(alac_amd.synth_pcm(3, n, f1), 1), (alac_amd.synth_pcm(4, n, f2), 2),
                                   (alac_amd.synth_pcm(5, n, f2), 2), (alac_amd.synth_pcm(6, n, f1), 1)], 16)
    enc = oracle.encoder(4096, 16, channels)
    stream, sizes = enc.encode_stream(pcm, n * 4096, 1)
    bare = bytes(enc.cookie())
    legacy = cont.cookie(0, bare)                      # 'frma' + 'alac' info atoms + terminator around the config
    from_mp4 = cont.parse_stsd(cont.build_stsd(bare, channels, 16, 44100))[0]
    assert from_mp4 == bare and len(legacy) == len(bare) + 32
    d_stream = torch.from_numpy(stream).cuda()
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])).cuda()
    for ck in (bare, legacy, legacy[12:], from_mp4):
        out, ns, st, f2 = gpu_ctx.decode(np.frombuffer(ck, np.uint8), d_stream, offs, n)
        gpu_ctx.synchronize()
        assert (f2.frame_size, f2.bit_depth, f2.num_channels, f2.sample_rate) == (4096, 16, channels, 44100)
        assert int(st.abs().sum()) == 0 and int(ns.sum()) == n * 4096
        assert np.array_equal(out.cpu().numpy(), pcm)


def _escape_overrun(pkt, depth, channels, frame):
    """is this a packet whose first element is UNCOMPRESSED and whose fixed-width payload ends past the packet?  (The reference
    reads such a payload past the end of its buffer and reports no error; this library refuses it: documented local hardening.)"""
    bits = np.unpackbits(np.ascontiguousarray(pkt, np.uint8))
    if bits.size < 23:
        return False
    val = lambda a, n: int("".join(map(str, bits[a:a + n])), 2) if a + n <= bits.size else 0  # noqa: E731
    tag = val(0, 3)
    if tag not in (0, 1, 3) or val(7, 12) != 0:
        return False
    hb = val(19, 4)
    if not (hb & 1):
        return False
    pos, n = 23, frame
    if hb >> 3:
        n = val(pos, 32)
        pos += 32
    ech = 2 if tag == 1 else 1
    return pos + n * ech * depth > bits.size


@pytest.mark.parametrize("block", range(4))
def test_corrupt_packets_differential(gpu_ctx, oracle, block):
    """truncated, bit-flipped, overwritten and over-long packets (640 per block, five depths, mono / stereo, six frame sizes),
    GPU against the oracle: nothing faults or hangs; wherever both accept a packet the PCM is identical; the ONLY status
    differences are packets with an uncompressed element cut short, which the reference decodes from whatever lies behind its
    buffer (codec/ALACDecoder.cu:697-727, :856-896 read without a bounds test) and this library refuses with kALAC_ParamError"""
    import torch
    both0 = 0
    for seed in range(block * 10, block * 10 + 10):
        rng = np.random.default_rng(90000 + seed)
        depth = int(rng.choice([16, 16, 24, 20, 32]))
        channels = int(rng.choice([1, 2, 2]))
        frame = int(rng.choice([4096, 1024, 512, 100, 64, 2048]))
        fmt = alac_amd.make_format(frame, depth, channels)
        n = 64
        pcm = alac_amd.synth_pcm(seed * 64, n, fmt)
        enc = oracle.encoder(frame, depth, channels)
        pk = []
        for p in range(n):
            enc.reset()
            a = enc.encode_packet(pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes], frame).copy()
            kind = int(rng.integers(0, 6))
            if kind == 0:
                a = a[:int(rng.integers(1, len(a) + 1))]
            elif kind in (1, 2):
                for _ in range(int(rng.integers(1, 4))):
                    a[int(rng.integers(0, len(a)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 3:
                a[int(rng.integers(0, min(len(a), 24)))] = int(rng.integers(0, 256))
            elif kind == 4:
                a = np.concatenate([a, rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)])
            pk.append(a)
        ck = enc.cookie()
        stream = np.concatenate(pk)
        offs = np.concatenate([[0], np.cumsum([len(x) for x in pk])]).astype(np.int64)
        out, ns, st, _ = gpu_ctx.decode(ck, torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), n)
        gpu_ctx.synchronize()
        out, ns, st = out.cpu().numpy(), ns.cpu().numpy(), st.cpu().numpy()
        dec = oracle.decoder(ck)
        bpf = fmt.bytes_per_frame
        for p in range(n):
            ost, want, m = dec.decode_packet(pk[p], bpf)
            assert st[p] in (0, -50, -4), (seed, p, int(st[p]))
            if (ost == 0) != (st[p] == 0):
                assert ost == 0 and st[p] == -50 and _escape_overrun(pk[p], depth, channels, frame), (seed, p, ost, int(st[p]), len(pk[p]))
            elif ost == 0:
                both0 += 1
                assert ns[p] == m and np.array_equal(out[p * fmt.packet_bytes:p * fmt.packet_bytes + m * bpf], want), (seed, p)
    assert both0 > 300


@pytest.mark.parametrize("depth", [16, 24])
@pytest.mark.parametrize("separate", [False, True])
def test_stream_at_any_byte_alignment(gpu_ctx, oracle, depth, separate):
    """The caller's stream may start at any byte: the header window (LDS copy of a packet's first 64 bytes, dword loads), the
    direct reads of the separate launches and the escape-element groups all assume a dword-aligned base and must fall back to
    their byte / staged forms otherwise.  The same packets (compressed, escaped, partial, one within 68 bytes of the stream's
    end) decode identically from a base shifted by 0..3 bytes, in the fused launch and in the separate launches."""
    import torch
    fmt = alac_amd.make_format(4096, depth, 2)
    n = 96
    rng = np.random.default_rng(77 + depth)
    pcm = alac_amd.synth_pcm(5, n, fmt).copy()
    noise = rng.integers(0, 256, fmt.packet_bytes, dtype=np.uint8)
    for p in range(3, n, 7):  # full-scale noise: escape packets
        pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes] = noise
    sizes = [4096] * n
    sizes[10], sizes[11], sizes[n - 1] = 100, 4095, 4  # partial frames; a tiny last packet near the stream's end
    enc = oracle.encoder(4096, depth, 2)
    pkts = []
    for p in range(n):
        enc.reset()
        pkts.append(enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + sizes[p] * fmt.bytes_per_frame], sizes[p]))
    assert any(len(pkts[p]) > fmt.packet_bytes for p in range(3, n, 7))  # escapes really happened
    stream = np.concatenate(pkts)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in pkts])]).astype(np.int64)
    cookie = oracle.encoder(4096, depth, 2).cookie()
    keep_fused = gpu_ctx.get_option("dec_fused")  # (a variant session pins these: put them back as they were)
    gpu_ctx.set_option("dec_fused", 0 if separate else 1)
    keep_direct = gpu_ctx.get_option("dec_direct")
    if separate and keep_direct == 1:
        gpu_ctx.set_option("dec_direct", 2)  # the direct reads at this batch size (automatic: from 80 000 packets on)
    try:
        ref = None
        for shift in range(4):
            buf = torch.zeros(len(stream) + 8, dtype=torch.uint8, device="cuda")
            buf[shift:shift + len(stream)] = torch.from_numpy(stream).cuda()
            view = buf[shift:shift + len(stream)]
            assert view.data_ptr() % 4 == shift
            out, ns, st, _ = gpu_ctx.decode(cookie, view, torch.from_numpy(offs).cuda(), n)
            gpu_ctx.synchronize()
            assert st.cpu().tolist() == [0] * n and ns.cpu().tolist() == sizes, shift
            out = out.cpu().numpy()
            for p in range(n):
                a, k = p * fmt.packet_bytes, sizes[p] * fmt.bytes_per_frame
                assert np.array_equal(out[a:a + k], pcm[a:a + k]), (shift, p)
            if ref is None:
                ref = out
            assert np.array_equal(out, ref), shift
    finally:
        gpu_ctx.set_option("dec_fused", keep_fused)
        gpu_ctx.set_option("dec_direct", keep_direct)
