"""CPU suite: the decoder restatement (oracle/alac_oracle.c `oalac_decode_packet`) on packets with FOREIGN header and
cookie parameters — what another legal ALAC encoder emits (codec/ALACDecoder.cu:795-857) — forged by oracle/forge.py.
Pins: tests/golden/forged.npz (forged and decoded over the reference's compiled stage objects) and, where oracle/_ref
is present, the same comparison live."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import forge  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_forged():
    z = np.load(os.path.join(GOLD, "forged.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def test_forged_golden_decodes_to_reference_pcm(oracle):
    z, meta = load_forged()
    assert sum(m["packets"] for m in meta) >= 150
    for m in meta:
        si = m["id"]
        dec = oracle.decoder(z[f"s{si}_cookie"])
        bpf = m["channels"] * forge.BPS[m["depth"]]
        stream, sizes, want = z[f"s{si}_stream"], z[f"s{si}_sizes"], z[f"s{si}_pcm"]
        off = woff = 0
        for sz in sizes:
            st, out, n = dec.decode_packet(stream[off:off + sz], bpf)
            assert st == 0, (m, st)
            assert np.array_equal(out, want[woff:woff + n * bpf]), m
            off += int(sz)
            woff += n * bpf
        assert woff == len(want)


def test_forger_is_reproducible_against_golden(oracle):
    """the forger over the ORACLE's stage functions writes the bytes the fixture holds (made over the reference's)"""
    from golden.make_golden import FORGED_STREAMS
    z, meta = load_forged()
    f = forge.Forger(oracle)
    for si, (depth, ch, frame, pb, mb, kb, count) in enumerate(FORGED_STREAMS):
        pk, pcm, ok = forge.forge_batch(f, np.random.default_rng(4000 + si), count, depth, ch, frame, pb, mb, kb)
        assert np.array_equal(np.concatenate(pk), z[f"s{si}_stream"]), si
        assert ok == meta[si]["lossless"]


@pytest.mark.parametrize("depth,channels", [(16, 2), (16, 1), (24, 2), (24, 1), (20, 2), (32, 2), (32, 1)])
def test_forged_packets_live_against_reference_objects(oracle, ref, depth, channels):
    fr = forge.Forger(oracle, dict(pc_block=ref.lib.pc_block, dyn_comp=ref.lib.ref_dyn_comp_flat, put_bits=ref.lib.ref_put_bits))
    fo = forge.Forger(oracle)
    bpf = channels * forge.BPS[depth]
    for pb, mb, kb in ((40, 10, 14), (17, 3, 7), (255, 255, 16), (40, 10, 1)):
        seed = depth * 1000 + channels * 100 + pb
        pk, pcm, ok = forge.forge_batch(fo, np.random.default_rng(seed), 30, depth, channels, 200, pb, mb, kb)
        pk2, _, _ = forge.forge_batch(fr, np.random.default_rng(seed), 30, depth, channels, 200, pb, mb, kb)
        ck = forge.cookie(200, depth, channels, pb, mb, kb)
        d_own, d_ref = oracle.decoder(ck), oracle.decoder(ck, hooks=ref.hooks())
        for a, b, p, k in zip(pk, pk2, pcm, ok):
            assert np.array_equal(a, b)
            st, out, n = d_own.decode_packet(a, bpf)
            st2, out2, n2 = d_ref.decode_packet(a, bpf)
            assert st == st2 == 0 and n == n2 == len(p) // bpf
            assert np.array_equal(out, out2)
            if k:
                assert np.array_equal(out, p)


def test_unsupported_element_tags(oracle):
    """ID_CCE / ID_PCE: kALAC_ParamError (codec/ALACDecoder.cu:932-939)"""
    f = forge.Forger(oracle)
    dec = oracle.decoder(forge.cookie(256, 16, 2))
    for tag in (2, 5):
        st, _, _ = dec.decode_packet(f.raw_tag(tag), 4)
        assert st == -50
