"""GPU parity, decode direction, FOREIGN streams: packets another legal ALAC encoder could emit — any numU / numV 0..31,
denShift != 9, pbFactor != 4, mode != 0 (first-order pass first), any mixBits / signed mixRes, shift-off bytes, partial
frames, LFE tags, cookies with other pb / mb / kb (codec/ALACDecoder.cu:795-857) — forged by oracle/forge.py and mixed in ONE
batch with packets shaped like this library's own encoder's, so that the fast predictor path, the generic predictor
(k_dec_unpc) and the lane fallback run side by side.  GPU decode == oracle decode (== source PCM where the forger
guarantees losslessness); the committed fixture tests/golden/forged.npz carries the reference objects' own answers."""
import json
import os
import sys

import numpy as np
import pytest

import alac_amd

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import forge  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpu_decode(ctx, cookie, packets):
    import torch
    stream = np.concatenate(packets)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in packets])]).astype(np.int64)
    out, ns, st, fmt = ctx.decode(cookie, torch.from_numpy(stream).cuda(), torch.from_numpy(offs).cuda(), len(packets))
    ctx.synchronize()
    return out.cpu().numpy(), ns.cpu().numpy(), st.cpu().numpy(), fmt


def test_forged_golden_fixture(gpu_ctx):
    """the reference objects' own answers (no oracle in the loop): GPU decode of the committed foreign packets"""
    z = np.load(os.path.join(GOLD, "forged.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    for m in meta:
        si = m["id"]
        sizes = z[f"s{si}_sizes"]
        ends = np.cumsum(sizes)
        stream = z[f"s{si}_stream"]
        pk = [stream[e - s:e] for s, e in zip(sizes, ends)]
        out, ns, st, fmt = gpu_decode(gpu_ctx, z[f"s{si}_cookie"], pk)
        assert st.tolist() == [0] * len(pk), (m, st.tolist())
        want, woff = z[f"s{si}_pcm"], 0
        for p in range(len(pk)):
            nb = int(ns[p]) * fmt.bytes_per_frame
            a = p * fmt.packet_bytes
            assert np.array_equal(out[a:a + nb], want[woff:woff + nb]), (m["depth"], m["channels"], p)
            woff += nb
        assert woff == len(want)


CASES = [  # depth, channels, frame, cookie (pb, mb, kb), forged packets
    (16, 2, 512, (40, 10, 14), 360), (16, 2, 4096, (40, 10, 14), 64), (16, 2, 512, (20, 5, 9), 120),
    (16, 1, 512, (40, 10, 14), 320), (16, 1, 384, (63, 30, 16), 96),
    (24, 2, 256, (40, 10, 14), 320), (24, 2, 256, (255, 255, 8), 64), (24, 1, 256, (30, 12, 11), 128),
    (20, 2, 256, (40, 10, 14), 160), (20, 1, 200, (50, 8, 13), 96),
    (32, 2, 128, (40, 10, 14), 160), (32, 1, 128, (25, 10, 12), 96),
]


@pytest.mark.parametrize("depth,channels,frame,agp,count", CASES)
def test_foreign_packets_mixed_with_own(gpu_ctx, oracle, depth, channels, frame, agp, count):
    pb, mb, kb = agp
    rng = np.random.default_rng(depth * 131 + channels * 17 + frame + pb)
    f = forge.Forger(oracle)
    pk, pcm, ok = forge.forge_batch(f, rng, count, depth, channels, frame, pb, mb, kb)
    kinds = [str(i) for i in ok.info]
    if agp == (40, 10, 14):
        # ordinary packets of the oracle's ENCODER (what this library writes) interleaved with the foreign ones
        fmt = alac_amd.make_format(frame, depth, channels)
        n_own = count // 3
        own = alac_amd.synth_pcm(5, n_own, fmt)
        enc = oracle.encoder(frame, depth, channels)
        for i in range(n_own):
            enc.reset()
            src = own[i * fmt.packet_bytes:(i + 1) * fmt.packet_bytes]
            at = int(rng.integers(0, len(pk) + 1))
            pk.insert(at, enc.encode_packet(src, frame))
            pcm.insert(at, src)
            ok.insert(at, True)
            kinds.insert(at, "own")
    ck = forge.cookie(frame, depth, channels, pb, mb, kb)
    out, ns, st, fmt = gpu_decode(gpu_ctx, ck, pk)
    dec = oracle.decoder(ck)
    bpf = fmt.bytes_per_frame
    bad = []
    for p, (a, src, k) in enumerate(zip(pk, pcm, ok)):
        ost, want, n = dec.decode_packet(a, bpf)
        assert ost == 0 and n * bpf == len(src)
        got = out[p * fmt.packet_bytes:p * fmt.packet_bytes + n * bpf]
        if st[p] != 0 or ns[p] != n or not np.array_equal(got, want):
            bad.append((p, kinds[p], int(st[p]), int(ns[p]), n))
        if k:
            assert np.array_equal(want, src), ("forger/oracle round trip", p)
    assert not bad, bad[:10]


@pytest.mark.parametrize("depth", [16, 24, 20])
def test_foreign_packets_in_the_separate_launch_regime(gpu_ctx, oracle, depth):
    """35 000 stereo packets (70 000 chains: work lists, one-lane predictor, pairs — 16-bit words and 20- / 24-bit six-byte
    frames with their shifted-off bytes) of which every 50th is foreign"""
    import torch
    frame, n = 128, 35000
    fmt = alac_amd.make_format(frame, depth, 2)
    bpf = fmt.bytes_per_frame
    d_pcm = gpu_ctx.synth_pcm(0, n, fmt)
    b = gpu_ctx.encode(fmt, d_pcm, n)
    gpu_ctx.synchronize()
    offs = b["offsets"].cpu().numpy()
    stream = b["out"][:int(offs[-1])].cpu().numpy()
    pcm = d_pcm.cpu().numpy()
    pk = [stream[offs[p]:offs[p + 1]] for p in range(n)]
    rng = np.random.default_rng(99 + depth)
    f = forge.Forger(oracle)
    fpk, fpcm, fok = forge.forge_batch(f, rng, n // 50, depth, 2, frame)
    where = list(range(7, n, 50))[:len(fpk)]
    for w, a in zip(where, fpk):
        pk[w] = a
    ck = forge.cookie(frame, depth, 2)
    out, ns, st, _ = gpu_decode(gpu_ctx, ck, pk)
    assert int(np.abs(st).sum()) == 0
    dec = oracle.decoder(ck)
    for w, a in zip(where, fpk):
        ost, want, m = dec.decode_packet(a, bpf)
        assert ost == 0 and ns[w] == m
        assert np.array_equal(out[w * fmt.packet_bytes:w * fmt.packet_bytes + m * bpf], want), (w, fok.info[where.index(w)])
    keep = np.ones(n, bool)
    keep[where] = False
    got = out.reshape(n, fmt.packet_bytes)[keep]
    assert np.array_equal(got, pcm.reshape(n, fmt.packet_bytes)[keep])
    assert (ns[keep] == frame).all()


def test_unsupported_and_skipped_elements(gpu_ctx, oracle):
    """ID_CCE / ID_PCE -> kALAC_ParamError (codec/ALACDecoder.cu:932-939) next to good packets"""
    f = forge.Forger(oracle)
    rng = np.random.default_rng(5)
    pk, pcm, ok = forge.forge_batch(f, rng, 6, 16, 2, 256)
    pk = [pk[0], f.raw_tag(2), pk[1], f.raw_tag(5), pk[2]]
    ck = forge.cookie(256, 16, 2)
    out, ns, st, fmt = gpu_decode(gpu_ctx, ck, pk)
    assert st.tolist() == [0, -50, 0, -50, 0]
    dec = oracle.decoder(ck)
    for p in (0, 2, 4):
        ost, want, n = dec.decode_packet(pk[p], 4)
        assert np.array_equal(out[p * fmt.packet_bytes:p * fmt.packet_bytes + n * 4], want)


@pytest.mark.parametrize("seed", list(range(24)) + [289])  # 289: found by the soak (32-bit mono, 8 taps, denShift 4: a wrapped walk)
def test_random_foreign_streams(gpu_ctx, oracle, seed):
    """randomised: depth, channels, frame size (odd ones too), cookie parameters and 40-90 forged packets per case, decoded in
    one call — GPU == oracle (== source where the forger guarantees losslessness).  tools/fuzz_soak.py runs more seeds."""
    rng = np.random.default_rng(7000 + seed)
    depth = int(rng.choice([16, 16, 24, 20, 32]))
    channels = int(rng.choice([1, 2, 2]))
    frame = int(rng.choice([4096, 1024, 512, 100, 333, 64, 17, 2048, 8, 24]))
    pb, mb, kb = int(rng.choice([40, 40, 20, 63, 255, 1])), int(rng.choice([10, 10, 1, 30, 255])), int(rng.choice([14, 14, 1, 8, 16]))
    count = int(rng.integers(40, 91))
    pk, pcm, ok = forge.forge_batch(forge.Forger(oracle), rng, count, depth, channels, frame, pb, mb, kb)
    ck = forge.cookie(frame, depth, channels, pb, mb, kb)
    out, ns, st, fmt = gpu_decode(gpu_ctx, ck, pk)
    dec = oracle.decoder(ck)
    bpf = fmt.bytes_per_frame
    for p, (a, src, k) in enumerate(zip(pk, pcm, ok)):
        ost, want, n = dec.decode_packet(a, bpf)
        assert ost == 0 and st[p] == 0 and ns[p] == n, (seed, p, ok.info[p], int(st[p]))
        assert np.array_equal(out[p * fmt.packet_bytes:p * fmt.packet_bytes + n * bpf], want), (seed, p, ok.info[p])
        if k:
            assert np.array_equal(want, src), (seed, p, "forger / oracle round trip")


@pytest.mark.parametrize("depth,channels", [(16, 2), (16, 1), (24, 2)])
def test_a_whole_batch_from_another_encoder(gpu_ctx, oracle, depth, channels):
    """6 000 packets ALL shaped like another encoder's: mode 0, prediction orders 1..8 and a denominator shift per packet (what
    ffmpeg's encoder writes: orders 4..6, a shift per frame), pbFactor 4 — through the separate launches (dec_fused = 0: the
    regime of large batches), where such chains take the one-lane predictor for any tap count up to 8 (unpc_any_body); every
    packet against the source, a sample against the oracle"""
    frame, n = 128, 6000
    rng = np.random.default_rng(31 + depth + channels)
    f = forge.Forger(oracle)
    pk, src = [], []
    for i in range(n):
        kind = int(rng.integers(0, 5))
        pcm = forge.test_signal(rng, kind, frame, depth, channels, headroom_bits=1)
        params = []
        for _ in range(channels):
            num, den = int(rng.choice([1, 2, 3, 4, 5, 6, 6, 7, 8])), int(rng.choice([4, 5, 6, 7, 8, 9, 9, 10]))
            cp = forge.ChannelParams(num, den, 4, 0)
            cp.coefs[:num] = rng.integers(-(1 << den), (1 << den) + 1, size=num) if i % 3 else forge.default_coefs(num, den)[:num]
            params.append(cp)
        shifted = 1 if depth == 24 else 0
        pk.append(f.element(pcm, frame, depth, channels, frame, params, mix_bits=2, mix_res=int(rng.integers(0, 5)) if channels == 2 else 0,
                            bytes_shifted=shifted))
        src.append(pcm)
    ck = forge.cookie(frame, depth, channels)
    with gpu_ctx.options(dec_fused=0):
        out, ns, st, fmt = gpu_decode(gpu_ctx, ck, pk)
    assert int(np.abs(st).sum()) == 0 and (ns == frame).all()
    got = out.reshape(n, fmt.packet_bytes)
    want = np.stack([np.frombuffer(bytes(x), np.uint8) for x in src])
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (bad[:8].tolist())
    dec = oracle.decoder(ck)
    for p in range(0, n, 499):
        ost, w, m = dec.decode_packet(pk[p], fmt.bytes_per_frame)
        assert ost == 0 and np.array_equal(w, want[p])
