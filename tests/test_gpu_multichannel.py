"""GPU parity for streams of 3..8 channels: the element sequence of sChannelMaps through the
C-ABI, bit-exact against the CPU oracle's element loop; decode through the lane decoder's element loop."""
import numpy as np
import pytest

import alac_amd
from oracle_lib import channel_elements, interleave_channels

pytestmark = pytest.mark.gpu


def _pcm(oracle, channels, depth, first, packets):
    parts = []
    for k, (ci, n) in enumerate(channel_elements(oracle, channels)):
        parts.append((alac_amd.synth_pcm(first + 16 * k, packets, alac_amd.make_format(4096, depth, n)), n))
    return interleave_channels(parts, depth)


@pytest.mark.parametrize("channels,depth", [(3, 16), (4, 24), (5, 16), (6, 16), (6, 24), (7, 20), (8, 16), (8, 32)])
def test_independent_packets_and_round_trip(gpu_ctx, oracle, channels, depth):
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    n = 20
    sizes_in = [4096] * n
    sizes_in[3], sizes_in[7], sizes_in[19] = 1000, 5, 404
    pcm = _pcm(oracle, channels, depth, 0, n)
    ns = torch.tensor(sizes_in, dtype=torch.int32).cuda()
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, num_samples=ns)
    enc = oracle.encoder(4096, depth, channels)
    off = 0
    for p, N in enumerate(sizes_in):
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + N * fmt.bytes_per_frame], N)
        assert sizes[p] == len(pk), (p, N)
        assert np.array_equal(stream[off:off + len(pk)], pk), (p, N)
        off += len(pk)
    assert off == len(stream)
    cookie = gpu_ctx.magic_cookie(fmt)
    assert np.array_equal(cookie, oracle.encoder(4096, depth, channels).cookie())  # fetched before encoding
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    out, nso, st, f2 = gpu_ctx.decode(cookie, torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), n)
    gpu_ctx.synchronize()
    assert f2.num_channels == channels and not st.cpu().numpy().any()
    assert np.array_equal(nso.cpu().numpy(), np.array(sizes_in))
    out = out.cpu().numpy()
    for p, N in enumerate(sizes_in):
        a = p * fmt.packet_bytes
        assert np.array_equal(out[a:a + N * fmt.bytes_per_frame], pcm[a:a + N * fmt.bytes_per_frame]), p


@pytest.mark.parametrize("channels,depth", [(6, 16), (3, 24)])
def test_chained_segments(gpu_ctx, oracle, channels, depth):
    """every element keeps its own coefficient rows across the packets of a segment; the state blocks are laid out
    [element][segment][64] and continue a chain in a second call"""
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    seg_first = [0, 3, 4, 9]
    nseg, n = len(seg_first) - 1, seg_first[-1]
    nel = len(channel_elements(oracle, channels))
    assert gpu_ctx.lib.alac_hip_state_int16(fmt) == 64 * nel
    pcm = _pcm(oracle, channels, depth, 7, n)
    sf = torch.tensor(seg_first, dtype=torch.int32).cuda()
    state = torch.zeros((nel, nseg, 64), dtype=torch.int16).cuda()
    half = [0, 2, 3, 6]  # first call: the first packets of every segment; second call: the rest, from the state
    idx1 = [p for s in range(nseg) for p in range(seg_first[s], seg_first[s] + half[s + 1] - half[s])]
    idx2 = [p for p in range(n) if p not in idx1]

    def take(idx):
        return np.concatenate([pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes] for p in idx])

    s1, z1 = gpu_ctx.encode_to_host(fmt, torch.from_numpy(take(idx1)).cuda(), len(idx1),
                                    seg_first=torch.tensor(half, dtype=torch.int32).cuda(), state=state)
    rest = [0]
    for s in range(nseg):
        rest.append(rest[-1] + (seg_first[s + 1] - seg_first[s]) - (half[s + 1] - half[s]))
    s2, z2 = gpu_ctx.encode_to_host(fmt, torch.from_numpy(take(idx2)).cuda(), len(idx2),
                                    seg_first=torch.tensor(rest, dtype=torch.int32).cuda(), state=state, state_in=True)
    whole, zw = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, seg_first=sf)
    got = {}
    o = 0
    for k, p in enumerate(idx1):
        got[p] = s1[o:o + int(z1[k])]
        o += int(z1[k])
    o = 0
    for k, p in enumerate(idx2):
        got[p] = s2[o:o + int(z2[k])]
        o += int(z2[k])
    off = 0
    for s in range(nseg):
        a, b = seg_first[s], seg_first[s + 1]
        ref, rs = oracle.encoder(4096, depth, channels).encode_stream(pcm[a * fmt.packet_bytes:b * fmt.packet_bytes],
                                                                      (b - a) * 4096, 0)
        assert np.array_equal(zw[a:b], rs)
        assert np.array_equal(whole[off:off + len(ref)], ref), s
        assert np.array_equal(np.concatenate([got[p] for p in range(a, b)]), ref), s
        off += len(ref)


def test_foreign_element_sequences_and_short_packets(gpu_ctx, oracle):
    """the decoder follows whatever elements a packet carries (codec/ALACDecoder.cu:600-990), not only the sequence
    this encoder writes: CPE+SCE and SCE+SCE+SCE for a 3-channel cookie (served by the lane decoder after the
    element rounds report the mismatch), a packet that ends after its first element (remaining channels zero),
    an LFE tag, and a truncated packet (status -50)"""
    import torch
    from oracle_lib import splice_elements
    depth, n = 16, 4
    f1, f2, f3 = (alac_amd.make_format(4096, depth, c) for c in (1, 2, 3))
    mono = [alac_amd.synth_pcm(3 + 8 * k, n, f1) for k in range(3)]
    stereo = alac_amd.synth_pcm(5, n, f2)

    def packets(pcm, ch):
        enc = oracle.encoder(4096, depth, ch)
        s, z = enc.encode_stream(pcm, n * 4096, 1)
        o = np.concatenate([[0], np.cumsum(z)]).astype(np.int64)
        return [s[o[p]:o[p + 1]] for p in range(n)]

    pm = [packets(m, 1) for m in mono]
    ps = packets(stereo, 2)
    cookie = oracle.encoder(4096, depth, 3).cookie()
    dec = oracle.decoder(cookie)

    def lfe(pk):  # ID_LFE (3) instead of ID_SCE (0) in the first 3 bits
        q = pk.copy()
        q[0] |= 0x60
        return q

    cases = {
        "cpe_sce": [splice_elements([(ps[p], 0), (pm[0][p], 0)]) for p in range(n)],
        "three_sce": [splice_elements([(pm[0][p], 0), (pm[1][p], 1), (pm[2][p], 2)]) for p in range(n)],
        "short": [splice_elements([(pm[0][p], 0)]) if p % 2 else splice_elements([(pm[0][p], 0), (ps[p], 0)])
                  for p in range(n)],
        "lfe_first": [splice_elements([(lfe(pm[0][p]), 0), (ps[p], 0)]) for p in range(n)],
    }
    trunc = [splice_elements([(pm[0][p], 0), (ps[p], 0)]) for p in range(n)]
    trunc[2] = trunc[2][:len(trunc[2]) // 2]
    cases["truncated"] = trunc
    for name, pks in cases.items():
        stream = np.concatenate(pks)
        offs = np.concatenate([[0], np.cumsum([len(q) for q in pks])]).astype(np.int64)
        out, ns, st, _ = gpu_ctx.decode(cookie, torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), n)
        gpu_ctx.synchronize()
        out, ns, st = out.cpu().numpy(), ns.cpu().numpy(), st.cpu().numpy()
        for p in range(n):
            want_st, want, want_n = dec.decode_packet(pks[p], f3.bytes_per_frame)
            assert st[p] == want_st, (name, p)
            if want_st == 0:
                assert ns[p] == want_n, (name, p)
                a = p * f3.packet_bytes
                assert np.array_equal(out[a:a + want_n * f3.bytes_per_frame], want), (name, p)


@pytest.mark.parametrize("channels,depth,frame", [(5, 24, 1000), (3, 16, 52), (7, 32, 4096 * 2), (4, 20, 333)])
def test_odd_frame_sizes(gpu_ctx, oracle, channels, depth, frame):
    """frame sizes that are no multiple of the tile / vector widths, incl. packets whose PCM is not 16-byte aligned"""
    import torch
    fmt = alac_amd.make_format(frame, depth, channels)
    n = 9
    rng = np.random.default_rng(frame + channels)
    bps = fmt.bytes_per_frame // channels
    t = np.arange(n * frame)
    cols = []
    for c in range(channels):
        amp = 1 << (depth - 3)
        x = (amp * np.sin(t * (0.002 + 0.01 * c)) + rng.integers(-amp // 16, amp // 16 + 1, t.size)).astype(np.int64)
        if depth == 20:
            x <<= 4
        b = (x & ((1 << (8 * bps)) - 1)).astype("<u8").view(np.uint8).reshape(-1, 8)[:, :bps]
        cols.append(b)
    pcm = np.ascontiguousarray(np.concatenate(cols, axis=1)).reshape(-1)
    sizes_in = [frame] * n
    sizes_in[2], sizes_in[8] = max(frame // 3, 1), max(frame - 1, 1)
    ns = torch.tensor(sizes_in, dtype=torch.int32).cuda()
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n, num_samples=ns)
    enc = oracle.encoder(frame, depth, channels)
    off = 0
    for p, N in enumerate(sizes_in):
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:p * fmt.packet_bytes + N * fmt.bytes_per_frame], N)
        assert sizes[p] == len(pk) and np.array_equal(stream[off:off + len(pk)], pk), (p, N)
        off += len(pk)
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    out, nso, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), n)
    gpu_ctx.synchronize()
    assert not st.cpu().numpy().any() and np.array_equal(nso.cpu().numpy(), np.array(sizes_in))
    out = out.cpu().numpy()
    for p, N in enumerate(sizes_in):
        a = p * fmt.packet_bytes
        assert np.array_equal(out[a:a + N * fmt.bytes_per_frame], pcm[a:a + N * fmt.bytes_per_frame]), p
