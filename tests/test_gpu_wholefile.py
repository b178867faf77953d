"""Whole-file CHAINED parity on the GPU (VERDICT r1 missing #5): the full sample data of the reference's
audio/50.wav (stereo, 237 packets) and audio/05.wav (mono, 302 packets) — committed as xz fixtures by
tests/golden/make_golden.py — encoded as ONE chained segment by the HIP path must have the size and FNV-1a-64 that
known_answers.json holds (produced by the encoder driver over the REFERENCE's own compiled pc_block / dyn_comp), and decode
back to the input."""
import json
import lzma
import os

import numpy as np
import pytest
import torch

import alac_amd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name,fixture", [("50.wav", "wav50_pcm.xz"), ("05.wav", "wav05_pcm.xz")])
def test_whole_file_chained_encode_matches_known_answers(gpu_ctx, oracle, name, fixture):
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        ka = json.load(f)["wav"][name]
    with open(os.path.join(HERE, "golden", fixture), "rb") as f:
        pcm = np.frombuffer(lzma.decompress(f.read()), np.uint8)
    ch, bits = ka["channels"], ka["bits"]
    fmt = alac_amd.make_format(4096, bits, ch, ka["rate"])
    total = pcm.size // fmt.bytes_per_frame
    assert total == ka["sample_frames"]
    stream, sizes, state = gpu_ctx.encode_host(fmt, pcm, total, segment_packets=0)  # 0 = one chained segment
    assert len(sizes) == ka["packets"]
    assert [int(x) for x in sizes[:8]] == ka["chained_first_sizes"] and int(sizes[-1]) == ka["chained_last_size"]
    assert stream.size == ka["chained_bytes"]
    assert f"{oracle.fnv(stream):016x}" == ka["chained_fnv"]
    # and back: the decoder never sees encoder state
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])).cuda()
    out, ns, st, _ = gpu_ctx.decode(gpu_ctx.magic_cookie(fmt), torch.from_numpy(stream).cuda(), offs, len(sizes))
    gpu_ctx.synchronize()
    assert int(st.abs().sum()) == 0 and int(ns.sum()) == total
    assert np.array_equal(out.cpu().numpy()[:pcm.size], pcm)


def test_whole_file_independent_packets_match_known_answers(gpu_ctx, oracle):
    """the same file with the state reset at every packet (BASELINE configs[1] semantics on real audio)"""
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        ka = json.load(f)["wav"]["50.wav"]
    with open(os.path.join(HERE, "golden", "wav50_pcm.xz"), "rb") as f:
        pcm = np.frombuffer(lzma.decompress(f.read()), np.uint8)
    fmt = alac_amd.make_format(4096, 16, 2, 44100)
    stream, sizes, _ = gpu_ctx.encode_host(fmt, pcm, ka["sample_frames"], segment_packets=1)
    assert stream.size == ka["indep_bytes"] and f"{oracle.fnv(stream):016x}" == ka["indep_fnv"]
