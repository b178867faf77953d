"""alacconvert (convert-utility/) on the GPU: files written by the product binary are byte-identical to what the
procedural oracle (oracle/caf_oracle.py over the C oracle codec) says the reference leaves on disk."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import caf_oracle as co  # noqa: E402
from container_lib import music_like  # noqa: E402
from test_container import oracle_codec  # noqa: E402

pytestmark = pytest.mark.gpu
CU = os.path.join(ROOT, "convert-utility")
BIN = os.path.join(CU, "alacconvert")


@pytest.fixture(scope="module")
def binary(gpu_ctx):
    subprocess.check_call(["make", "-C", CU, "alacconvert"], stdout=subprocess.DEVNULL)
    return BIN


def run(binary, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=e)
    return p.returncode, p.stdout, p.stderr


def oracle_encode(oracle, data, bits, ch, rate):
    cookie, enc, dec = oracle_codec(oracle, bits, ch, rate)
    return co.encode_file(data, cookie, enc), dec


def golden_pcm(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "packets.npz"))
    return z[name].tobytes()


@pytest.mark.parametrize("case", ["stereo16_music", "mono16_music", "stereo24", "stereo32", "stereo16_exact", "tiny"])
def test_single_file_round_trip_is_reference_identical(binary, oracle, tmp_path, case):
    if case == "stereo16_music":
        bits, ch, pcm = 16, 2, golden_pcm("stereo_pcm")          # excerpt of the reference's audio/50.wav
    elif case == "mono16_music":
        bits, ch, pcm = 16, 1, golden_pcm("mono_pcm")
    elif case == "stereo24":
        bits, ch, pcm = 24, 2, music_like(4096 * 2 + 77, 2, 24, 3)
    elif case == "stereo32":
        bits, ch, pcm = 32, 2, music_like(4096 + 5, 2, 32, 4)
    elif case == "stereo16_exact":
        bits, ch, pcm = 16, 2, music_like(4096 * 3, 2, 16, 5)    # phantom packet in the table header
    else:
        bits, ch, pcm = 16, 2, music_like(40, 2, 16, 6)
    wav = co.make_wav(pcm, ch, 44100, bits, extra_chunks=[(b"LIST", b"infoinfo")])
    src, caf, back = tmp_path / "in.wav", tmp_path / "out.caf", tmp_path / "back.wav"
    src.write_bytes(wav)
    rc, out, err = run(binary, src, caf)
    assert rc == 0, err
    assert out == f"Input file: {src}\nOutput file: {caf}\n"
    want, dec = oracle_encode(oracle, wav, bits, ch, 44100)
    got = caf.read_bytes()
    assert got == want
    rc, _, err = run(binary, caf, back)
    assert rc == 0, err
    assert back.read_bytes() == co.decode_file(want, True, dec)
    assert back.read_bytes()[44:] == pcm
    # CAF PCM out as well (any extension but .wav), and that file is a valid encode input again
    back_caf = tmp_path / "back.caf"
    assert run(binary, caf, back_caf)[0] == 0
    assert back_caf.read_bytes() == co.decode_file(want, False, dec)
    again = tmp_path / "again.caf"
    assert run(binary, back_caf, again)[0] == 0
    assert again.read_bytes() == want


def test_batch_outputs_equal_single_file_outputs(binary, oracle, tmp_path):
    specs = [(16, 2, 4096 * 4 + 100, 11), (16, 2, 4096 * 2, 12), (16, 1, 5000, 13), (24, 2, 4096 + 9, 14), (16, 2, 7, 15),
             (16, 2, 4096 * 7 + 1, 16)]
    args, wants = ["--batch"], []
    for i, (bits, ch, frames, seed) in enumerate(specs):
        pcm = music_like(frames, ch, bits, seed)
        wav = co.make_wav(pcm, ch, 48000 if i == 1 else 44100, bits)
        src, dst = tmp_path / f"in{i}.wav", tmp_path / f"out{i}.caf"
        src.write_bytes(wav)
        args += [src, dst]
        wants.append((dst, oracle_encode(oracle, wav, bits, ch, 48000 if i == 1 else 44100)[0], pcm))
    rc, _, err = run(binary, *args)
    assert rc == 0, err
    for dst, want, _ in wants:
        assert dst.read_bytes() == want
    # and the whole batch decodes in one call too
    args = ["--batch"]
    for i, (dst, _, _) in enumerate(wants):
        args += [dst, tmp_path / f"back{i}.wav"]
    rc, _, err = run(binary, *args)
    assert rc == 0, err
    for i, (_, _, pcm) in enumerate(wants):
        assert (tmp_path / f"back{i}.wav").read_bytes()[44:] == pcm


def test_batch_dealt_to_several_devices(binary, oracle, tmp_path):
    """--batch --devices N: the files go round-robin to N workers, one host thread and one context per device (replicas, no
    exchange).  A one-GPU box has one device: ALACCONVERT_SHARE_DEVICES=1 lets the three worker threads share it, which
    exercises the dealing and the threads; without it the tool refuses more devices than it sees.  Outputs are those of the
    single-device batch."""
    specs = [(16, 2, 4096 * 3 + 100, 41), (16, 1, 5000, 42), (24, 2, 4096 + 9, 43), (16, 2, 4096 * 5, 44), (16, 2, 33, 45),
             (16, 1, 4096 * 2 + 1, 46), (16, 2, 4096 * 2 + 7, 47)]
    args, wants = [], []
    for i, (bits, ch, frames, seed) in enumerate(specs):
        pcm = music_like(frames, ch, bits, seed)
        wav = co.make_wav(pcm, ch, 44100, bits)
        src, dst = tmp_path / f"in{i}.wav", tmp_path / f"out{i}.caf"
        src.write_bytes(wav)
        args += [src, dst]
        wants.append((dst, oracle_encode(oracle, wav, bits, ch, 44100)[0], pcm))
    rc, _, err = run(binary, "--batch", "--devices", 3, *args, env={"ALACCONVERT_SHARE_DEVICES": "1"})
    assert rc == 0, err
    for dst, want, _ in wants:
        assert dst.read_bytes() == want
    back = []
    for i, (dst, _, _) in enumerate(wants):
        back += [dst, tmp_path / f"back{i}.wav"]
    rc, _, err = run(binary, "--batch", "--devices", 2, *back, env={"ALACCONVERT_SHARE_DEVICES": "1"})
    assert rc == 0, err
    for i, (_, _, pcm) in enumerate(wants):
        assert (tmp_path / f"back{i}.wav").read_bytes()[44:] == pcm
    rc, _, err = run(binary, "--batch", "--devices", 1, *args)
    assert rc == 0, err
    import torch
    if torch.cuda.device_count() < 64:
        rc, _, err = run(binary, "--batch", "--devices", 64, *args)
        assert rc == 1 and "GPU(s) visible" in err
    rc, out, _ = run(binary, "--devices", 2, args[0], args[1])  # one file is one serial chain: nothing to deal out
    assert rc == 1 and out.startswith("Usage:")


@pytest.mark.parametrize("bits,ch,frames", [(16, 2, 4096 * 3 + 50), (24, 2, 4096 + 7), (16, 1, 9000), (16, 6, 4096 + 100)])
def test_m4a_out_and_in(binary, oracle, tmp_path, bits, ch, frames):
    """<out>.m4a on encode, an M4A input on decode.  The packets and the cookie inside the M4A are those of the CAF
    the same input gives (the reference-identical file), the box tables describe them (tests/test_m4a.py walks them
    independently), and the file decodes back to the input — alone and in a batch next to a CAF."""
    import struct
    from container_lib import Container
    pcm = music_like(frames, ch, bits, 60 + ch)
    wav = co.make_wav(pcm, ch, 44100, bits)
    src, m4a, caf = tmp_path / "in.wav", tmp_path / "out.m4a", tmp_path / "out.caf"
    src.write_bytes(wav)
    assert run(binary, src, m4a)[0] == 0 and run(binary, src, caf)[0] == 0
    ct = Container()
    info, cookie, sizes, pos = ct.parse_alac_m4a(m4a.read_bytes())
    ck2, sizes2, dpos = ct.parse_alac_caf(caf.read_bytes())
    assert cookie == ck2 and list(sizes) == list(sizes2)
    f, c = m4a.read_bytes(), caf.read_bytes()
    assert f[int(pos[0]):int(pos[0]) + int(sizes.sum())] == c[dpos:dpos + int(sizes2.sum())]
    i = f.index(b"mdhd")
    assert struct.unpack(">II", f[i + 16:i + 24]) == (44100, frames)
    if ch <= 2:
        back = tmp_path / "back.wav"
        rc, _, err = run(binary, m4a, back)
        assert rc == 0, err
        assert back.read_bytes()[44:] == pcm
        b1, b2 = tmp_path / "b1.wav", tmp_path / "b2.wav"
        rc, _, err = run(binary, "--batch", m4a, b1, caf, b2)
        assert rc == 0, err
        assert b1.read_bytes() == b2.read_bytes() == back.read_bytes()
    else:
        back = tmp_path / "back.caf"
        rc, _, err = run(binary, m4a, back)
        assert rc == 0, err
        assert back.read_bytes().endswith(pcm)
    again = tmp_path / "again.mp4"  # PCM CAF / WAV in, .mp4 out: the same container
    assert run(binary, back, again)[0] == 0
    assert again.read_bytes() == f


def test_segment_mode_is_valid_alac_and_decodes_to_the_input(binary, oracle, tmp_path):
    bits, ch = 16, 2
    pcm = music_like(4096 * 6 + 300, ch, bits, 21)
    wav = co.make_wav(pcm, ch, 44100, bits)
    src, caf = tmp_path / "in.wav", tmp_path / "seg.caf"
    src.write_bytes(wav)
    rc, _, err = run(binary, "--segment-packets", 2, src, caf)
    assert rc == 0, err
    cookie, _, dec = oracle_codec(oracle, bits, ch, 44100)
    got = caf.read_bytes()
    assert co.decode_file(got, True, dec)[44:] == pcm       # any ALAC decoder accepts it
    # segments of 2 packets: packets 0-1 equal the chained encode, the stream as a whole does not
    enc = oracle.encoder(4096, bits, ch, 44100)
    stream, sizes = enc.encode_stream(np.frombuffer(pcm, np.uint8), len(pcm) // 4, segment_packets=2)
    dpos = got.index(b"data") + 16
    assert got[dpos:dpos + len(stream)] == stream.tobytes()


def test_bad_invocations(binary, tmp_path):
    assert run(binary)[0] == 1
    rc, out, _ = run(binary, "-x", "a", "b")
    assert rc == 1 and out.startswith("unknown option: -x\n")
    rc, _, err = run(binary, tmp_path / "missing.wav", tmp_path / "o.caf")
    assert rc == 1 and "Cannot open file" in err
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"not a wave file at all")
    rc, _, err = run(binary, junk, tmp_path / "o.caf")
    assert rc == 1 and "Cannot determine what format" in err


@pytest.mark.parametrize("bits,ch,frames", [(16, 6, 4096 * 2 + 300), (24, 3, 4096 + 11), (16, 8, 4096)])
def test_multichannel_files(binary, oracle, tmp_path, bits, ch, frames):
    """3..8 channels: 'chan' chunk after the 48-byte 'kuki', element packets; decode only to CAF
    (convert-utility/main.cu:169-174 refuses WAVE above two channels)"""
    pcm = music_like(frames, ch, bits, 30 + ch)
    wav = co.make_wav(pcm, ch, 48000, bits)
    src, caf, back = tmp_path / "in.wav", tmp_path / "out.caf", tmp_path / "back.caf"
    src.write_bytes(wav)
    rc, _, err = run(binary, src, caf)
    assert rc == 0, err
    cookie, enc, dec = oracle_codec(oracle, bits, ch, 48000)
    assert len(cookie) == 48
    want = co.encode_file(wav, cookie, enc)
    assert caf.read_bytes() == want
    rc, _, err = run(binary, caf, tmp_path / "back.wav")
    assert rc == 1 and "more than two channels" in err
    rc, _, err = run(binary, caf, back)
    assert rc == 0, err
    got = back.read_bytes()
    assert got == co.decode_file(want, False, dec)
    assert got.endswith(pcm)
    again = tmp_path / "again.caf"
    assert run(binary, back, again)[0] == 0
    assert again.read_bytes() == want


@pytest.mark.parametrize("ch,frames", [(2, 4096 * 2 + 333), (1, 4096 + 40)])
def test_20_bit_files(binary, oracle, tmp_path, ch, frames):
    """20-bit material lives in 3-byte containers, left-justified (what the codec's mix20 / copy20ToPredictor read).  The
    reference's utility sizes such a file with 20 >> 3 = 2 bytes per sample (convert-utility/main.cu:389) and so never handled
    one; here the file is sized by its containers: the packets are the ones the codec oracle makes of the same PCM, in CAF and
    in M4A, and both decode back to the input."""
    import struct
    from container_lib import Container
    rng = np.random.default_rng(20 + ch)
    t = np.arange(frames)
    cols = [np.round((0.3 * np.sin(2 * np.pi * (300.0 + 90 * c) * t / 44100.0) + 0.02 * rng.standard_normal(frames)) * ((1 << 19) - 1))
            for c in range(ch)]
    v = np.stack(cols, axis=1).astype(np.int64)
    pcm = ((v << 4) & 0xffffff).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    bpf = ch * 3
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, 48000, 48000 * bpf, bpf, 20) + b"data" + struct.pack("<I", len(pcm)) + pcm
    wav = b"RIFF" + struct.pack("<I", len(body)) + body
    src, caf, m4a = tmp_path / "in.wav", tmp_path / "out.caf", tmp_path / "out.m4a"
    src.write_bytes(wav)
    rc, _, err = run(binary, src, caf)
    assert rc == 0, err
    assert run(binary, src, m4a)[0] == 0
    ct = Container()
    cookie, sizes, dpos = ct.parse_alac_caf(caf.read_bytes())
    enc = oracle.encoder(4096, 20, ch, 48000)
    assert cookie == bytes(enc.cookie())
    want = []
    for p0 in range(0, frames, 4096):
        n = min(4096, frames - p0)
        want.append(bytes(enc.encode_packet(np.frombuffer(pcm[p0 * bpf:(p0 + n) * bpf], np.uint8), n)))
    assert [int(s) for s in sizes] == [len(w) for w in want]
    assert caf.read_bytes()[dpos:dpos + sum(len(w) for w in want)] == b"".join(want)
    info, cookie2, sizes2, pos2 = ct.parse_alac_m4a(m4a.read_bytes())
    assert cookie2 == cookie and list(sizes2) == list(sizes)
    for packed in (caf, m4a):
        back = tmp_path / (packed.name + ".wav")
        rc, _, err = run(binary, packed, back)
        assert rc == 0, err
        b = back.read_bytes()
        assert b[44:] == pcm
        assert struct.unpack("<HHIIHH", b[20:36]) == (1, ch, 48000, 48000 * bpf, bpf, 20)
    # PCM CAF out (3-byte samples, 20 valid bits) is a valid encode input again
    back_caf, again = tmp_path / "back.caf", tmp_path / "again.caf"
    assert run(binary, caf, back_caf)[0] == 0
    assert back_caf.read_bytes().endswith(pcm)
    assert run(binary, back_caf, again)[0] == 0
    assert again.read_bytes() == caf.read_bytes()
