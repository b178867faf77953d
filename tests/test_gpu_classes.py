"""The drop-in C++ classes driven with the reference tool's own call sequence (tests/cpp/fork_calls.hip mirrors
convert-utility/main.cu's EncodeALAC / DecodeALAC): device buffer -> InitializeSampling -> Encode(index), and
Decode(index) -> fillWriteBuffer.  Output must equal the oracle's chained encode and decode back to the input."""
import os
import subprocess

import numpy as np
import pytest

from container_lib import music_like

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


@pytest.fixture(scope="module")
def harness(gpu_ctx):
    subprocess.check_call(["make", "-C", CPP, "fork_calls"], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, "fork_calls")


@pytest.mark.parametrize("bits,ch,frames", [(16, 2, 4096 * 5 + 321), (16, 1, 4096 * 2 + 7), (24, 2, 4096 * 2), (32, 2, 4096 + 100),
                                             (16, 6, 4096 * 3 + 50), (24, 3, 4096 + 9)])
def test_reference_call_sequence(harness, oracle, tmp_path, bits, ch, frames):
    pcm = music_like(frames, ch, bits, seed=bits + ch)
    (tmp_path / "in.pcm").write_bytes(pcm)
    p = subprocess.run([harness, str(bits), str(ch), "44100", str(tmp_path / "in.pcm"), str(tmp_path / "s.bin"),
                        str(tmp_path / "z.bin"), str(tmp_path / "back.pcm")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    enc = oracle.encoder(4096, bits, ch, 44100)
    ref, ref_sizes = enc.encode_stream(np.frombuffer(pcm, np.uint8), frames, segment_packets=0)  # chained, as one file
    got = np.fromfile(tmp_path / "s.bin", np.uint8)
    got_sizes = np.fromfile(tmp_path / "z.bin", np.uint32)
    assert np.array_equal(got_sizes, ref_sizes)
    assert np.array_equal(got, ref)
    assert (tmp_path / "back.pcm").read_bytes() == pcm


def test_class_path_reports_a_lost_handoff(harness, tmp_path):
    """the drop-in ALACEncoder batch path (InitializeSampling -> Encode(index), what alacconvert uses) must fail
    when a consumer wave of an in-launch hand-off gave up, not hand out corrupt packets with ALAC_noErr.  The context the
    class creates takes ALAC_HIP_DEBUG_LOSE_HANDOFF as its default: InitializeSampling's LastStatus is the error
    (harness exit code 5)."""
    frames = 4096 * 5 + 321
    pcm = music_like(frames, 2, 16, seed=18)
    (tmp_path / "in.pcm").write_bytes(pcm)
    env = dict(os.environ, ALAC_HIP_DEBUG_LOSE_HANDOFF="1")
    p = subprocess.run([harness, "16", "2", "44100", str(tmp_path / "in.pcm"), str(tmp_path / "s.bin"),
                        str(tmp_path / "z.bin"), str(tmp_path / "back.pcm")], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 5, (p.returncode, p.stdout, p.stderr)
    assert not (tmp_path / "s.bin").exists()


@pytest.mark.parametrize("bits,ch", [(16, 2), (24, 2), (16, 6)])
def test_set_fast_mode_through_the_class(harness, oracle, tmp_path, bits, ch):
    """ALACEncoder::SetFastMode(true) (codec/ALACEncoder.h:44): the reference tool's call sequence with the search-free path;
    the chained stream equals the oracle's EncodeStereoFast restatement and decodes back through ALACDecoder"""
    frames = 4096 * 4 + 77
    pcm = music_like(frames, ch, bits, seed=bits + ch + 1)
    (tmp_path / "in.pcm").write_bytes(pcm)
    p = subprocess.run([harness, str(bits), str(ch), "44100", str(tmp_path / "in.pcm"), str(tmp_path / "s.bin"),
                        str(tmp_path / "z.bin"), str(tmp_path / "back.pcm"), "fast"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    enc = oracle.encoder(4096, bits, ch, 44100, fast=True)
    ref, ref_sizes = enc.encode_stream(np.frombuffer(pcm, np.uint8), frames, segment_packets=0)
    assert np.array_equal(np.fromfile(tmp_path / "z.bin", np.uint32), ref_sizes)
    assert np.array_equal(np.fromfile(tmp_path / "s.bin", np.uint8), ref)
    assert (tmp_path / "back.pcm").read_bytes() == pcm
