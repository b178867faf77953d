"""CPU tests of the alacconvert container code (convert-utility/container.cpp) against the procedural oracle
(oracle/caf_oracle.py) and the reference's known answers.  No GPU, no HIP."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import caf_oracle as co  # noqa: E402
from container_lib import Container, music_like  # noqa: E402

REF_AUDIO = "/root/reference/audio"


@pytest.fixture(scope="module")
def cont():
    return Container()


def oracle_codec(oracle, bits, ch, rate):
    enc = oracle.encoder(4096, bits, ch, rate)
    cookie = bytes(enc.cookie())
    dec = oracle.decoder(np.frombuffer(cookie, np.uint8))
    bpf = ch * (bits >> 3)

    def encode_packet(pcm, frames):
        return bytes(enc.encode_packet(np.frombuffer(pcm, np.uint8), frames))

    def decode_packet(pkt):
        st, pcm, ns = dec.decode_packet(np.frombuffer(pkt, np.uint8), bpf, 4096)
        assert st == 0
        return bytes(pcm), ns

    return cookie, encode_packet, decode_packet


def test_ber_known_answers_and_round_trip(cont):
    # CAFFileALAC.cpp:189-236: 7 bits per byte, MSB group first
    for v, want in [(0, b"\x00"), (127, b"\x7f"), (128, b"\x81\x00"), (5697, b"\xac\x41"), (16383, b"\xff\x7f"),
                    (16384, b"\x81\x80\x00"), (16392, b"\x81\x80\x08"), (2097151, b"\xff\xff\x7f"), (2097152, b"\x81\x80\x80\x00"),
                    (0x0fffffff, b"\xff\xff\xff\x7f"), (0x10000000, b"\x81\x80\x80\x80\x00")]:
        assert cont.ber(v) == want == co.ber(v)
        assert cont.read_ber(want + b"\xff\xff") == (v, len(want))
        assert co.read_ber(want + b"\xff\xff", 5) == (v, len(want))


@pytest.mark.parametrize("bits,ch,frames", [(16, 2, 4096 * 3 + 1000), (16, 2, 4096 * 2), (16, 1, 4096 + 17), (24, 2, 5000),
                                            (32, 2, 4096 + 1), (16, 2, 100), (24, 1, 4096 * 2)])
def test_encode_layout_matches_procedural_oracle(cont, oracle, bits, ch, frames):
    pcm = music_like(frames, ch, bits, seed=bits + ch + frames)
    wav = co.make_wav(pcm, ch, 44100, bits)
    cookie, encode_packet, decode_packet = oracle_codec(oracle, bits, ch, 44100)
    sizes, stream = [], b""

    def enc_and_keep(p, n):
        pkt = encode_packet(p, n)
        sizes.append(len(pkt))
        nonlocal stream
        stream += pkt
        return pkt

    want = co.encode_file(wav, cookie, enc_and_keep)
    rc, info, err = cont.sniff(wav)
    assert rc == 0, err
    assert (info.channels, info.bits_per_channel, info.data_size, info.is_alac) == (ch, bits, len(pcm), 0)
    assert wav[info.data_pos:info.data_pos + info.data_size] == pcm
    got = cont.build_alac_caf(44100.0, ch, bits, len(pcm), cookie, sizes, stream)
    assert got == want
    # the file parses back to the same packets, and decoding them gives the input
    ck, psz, dpos = cont.parse_alac_caf(got)
    assert ck == cookie and list(psz) == sizes and got[dpos:dpos + len(stream)] == stream
    back = co.decode_file(got, True, decode_packet)
    assert back == cont.build_wave(44100.0, ch, bits, pcm)
    assert back[44:] == pcm
    assert co.decode_file(got, False, decode_packet) == cont.build_wave(44100.0, ch, bits, pcm, caf=True)


def test_header_bytes_and_packet_table_quirks(cont, oracle):
    """caff header, desc, kuki size byte, pakt numbers incl. the phantom packet, entry widths, free chunk."""
    bits, ch = 16, 2
    frames = 4096 * 2  # exact multiple: the header counts 3 packets, remainder frames = 4096
    pcm = music_like(frames, ch, bits, seed=5)
    cookie, encode_packet, _ = oracle_codec(oracle, bits, ch, 44100)
    pk = [encode_packet(pcm[i * 16384:(i + 1) * 16384], 4096) for i in range(2)]
    f = cont.build_alac_caf(44100.0, ch, bits, len(pcm), cookie, [len(p) for p in pk], b"".join(pk))
    assert f[:8] == bytes.fromhex("6361666600010000")
    assert f[8:20] == b"desc" + b"\0" * 7 + b"\x20"
    rate, fid, flags, bpp, fpp, chans, bpc = struct.unpack(">d4sIIIII", f[20:52])
    assert (rate, fid, flags, bpp, fpp, chans, bpc) == (44100.0, b"alac", 1, 0, 4096, 2, 0)
    assert f[52:64] == b"kuki" + b"\0" * 7 + b"\x18" and f[64:88] == cookie
    assert cookie == bytes.fromhex("000010000010280a0e0200ff00000000000000000000ac44")
    assert f[88:92] == b"pakt"
    npk, valid, priming, remainder = struct.unpack(">qqii", f[100:124])
    assert (npk, valid, priming, remainder) == (3, 8192, 0, 4096)
    # 16-bit stereo: 3-byte entries reserved (max packet 16392 >= 16384), 2-byte BER written -> 9 - 4 = 5 left: no free chunk
    assert struct.unpack(">q", f[92:100])[0] == 9 + 24
    assert f[124 + 4:124 + 9] == b"\0" * 5 and f[133:137] == b"data"
    # mono: 2-byte entries (8200 < 16384)
    _, _, valid_m, _ = co.base_packet_table(16, 1, 4096 * 2 * 5 + 2)
    assert co.base_packet_table(16, 1, 4096 * 2 * 5 + 2)[0] == 2 * 6 and valid_m == 4096 * 5 + 1
    # more than 12 bytes left over -> 'free' chunk and an 8-byte patched pakt size
    frames = 4096 * 14 + 5
    pcm = music_like(frames, ch, bits, seed=6)
    cookie, encode_packet, _ = oracle_codec(oracle, bits, ch, 44100)
    pk = [encode_packet(pcm[i * 16384:(i + 1) * 16384], min(4096, frames - i * 4096)) for i in range(15)]
    f = cont.build_alac_caf(44100.0, ch, bits, len(pcm), cookie, [len(p) for p in pk], b"".join(pk))
    used = sum(len(co.ber(len(p))) for p in pk)
    assert 45 - used > 12
    assert struct.unpack(">q", f[92:100])[0] == used + 24
    fpos = 124 + used
    assert f[fpos:fpos + 4] == b"free" and struct.unpack(">q", f[fpos + 4:fpos + 12])[0] == 45 - used - 12
    assert f[fpos + 45 - used:fpos + 49 - used] == b"data"


def test_sniffers_agree_on_awkward_inputs(cont):
    pcm = music_like(300, 2, 16, seed=1)
    # chunks before 'fmt ' and between 'fmt ' and 'data' are skipped (main.cu:252-258, :357-361)
    wav = co.make_wav(pcm, 2, 48000, 16, extra_chunks=[(b"LIST", b"abcdefgh"), (b"junk", b"1234")])
    body = wav[12:]
    i = body.index(b"data")
    wav2 = wav[:12] + body[:i] + b"cue " + struct.pack("<I", 6) + b"zzzzzz" + body[i:]
    wav2 = wav2[:4] + struct.pack("<I", len(wav2) - 8) + wav2[8:]
    for w in (wav, wav2):
        rc, info, _ = cont.sniff(w)
        fmt = co.get_input_format(w)
        pos, size = co.find_data_start(w, "WAVE")
        assert rc == 0 and (info.data_pos, info.data_size) == (pos, size) == (w.index(b"data") + 8, len(pcm))
        assert (info.channels, info.bits_per_channel, info.sample_rate) == (fmt["channels"], fmt["bits"], fmt["rate"])
    # WAVE_FORMAT_EXTENSIBLE is refused by both (main.cu:229-234)
    ext = bytearray(wav)
    k = wav.index(b"fmt ") + 8
    ext[k:k + 2] = b"\xfe\xff"
    assert cont.sniff(bytes(ext))[0] != 0 and co.get_input_format(bytes(ext)) is None
    # CAF lpcm, big and little endian
    for little in (True, False):
        caf = co.make_pcm_caf(pcm, 2, 44100, 16, little_endian=little)
        rc, info, _ = cont.sniff(caf)
        fmt = co.get_input_format(caf)
        pos, size = co.find_data_start(caf, "caff")
        assert rc == 0 and info.is_alac == 0 and (info.data_pos, info.data_size) == (pos, size)
        assert bool(info.big_endian_pcm) == (not little) == ((fmt["flags"] & 2) != 0)
    assert cont.sniff(b"RIFFxxxxAVI LIST")[0] != 0
    assert cont.sniff(b"short")[0] != 0


def test_big_endian_caf_input_encodes_like_the_wav(cont, oracle):
    bits, ch, frames = 24, 2, 4096 + 333
    pcm = music_like(frames, ch, bits, seed=9)
    be = co.swap_to_little(pcm, bits)  # the same swap turns LE into BE
    outs = []
    for data in (co.make_wav(pcm, ch, 44100, bits), co.make_pcm_caf(be, ch, 44100, bits, little_endian=False)):
        cookie, encode_packet, _ = oracle_codec(oracle, bits, ch, 44100)
        outs.append(co.encode_file(data, cookie, encode_packet))
    assert outs[0] == outs[1]


@pytest.mark.skipif(not os.path.isdir(REF_AUDIO), reason="reference audio only exists in the build container")
def test_reference_wav_known_answers(cont, oracle):
    """audio/50.wav -> 237 packets, 1 164 578 payload bytes; the CAF decodes back to the WAV's data chunk."""
    with open(os.path.join(REF_AUDIO, "50.wav"), "rb") as fh:
        wav = fh.read()
    fmt = co.get_input_format(wav)
    cookie, encode_packet, decode_packet = oracle_codec(oracle, fmt["bits"], fmt["channels"], int(fmt["rate"]))
    sizes = []

    def enc(p, n):
        pkt = encode_packet(p, n)
        sizes.append(len(pkt))
        return pkt

    caf = co.encode_file(wav, cookie, enc)
    assert len(sizes) == 237 and sum(sizes) == 1164578
    rc, info, _ = cont.sniff(wav)
    ck, psz, dpos = cont.parse_alac_caf(caf)
    assert list(psz) == sizes and ck == cookie
    got = cont.build_alac_caf(fmt["rate"], fmt["channels"], fmt["bits"], info.data_size, cookie, sizes,
                              caf[dpos:dpos + sum(sizes)])
    assert got == caf
    back = co.decode_file(caf, True, decode_packet)
    assert back[44:] == wav[info.data_pos:info.data_pos + info.data_size]


def test_cookie_outside_caf(cont, oracle):
    """legacy 'frma' / 'alac' wrapping and the MP4 sample description of ALACMagicCookieDescription.txt:177-238, byte
    for byte; the decoders (oracle restating codec/ALACDecoder.cu:123-134) accept the wrapped cookie"""
    import struct
    for ch in (2, 6):
        enc = oracle.encoder(4096, 16, ch, 44100)
        ck = bytes(enc.cookie())
        assert len(ck) == (24 if ch == 2 else 48)
        w = cont.cookie(0, ck)
        assert w == (struct.pack(">I4s4s", 12, b"frma", b"alac") + struct.pack(">I4sI", 12 + len(ck), b"alac", 0) + ck +
                     struct.pack(">II", 8, 0))
        assert cont.cookie(1, w) == ck and cont.cookie(1, ck) == ck
        assert cont.cookie(1, w[12:]) == ck            # 'alac' info without the format atom
        assert cont.cookie(1, b"\\0" * 10) == b""
        d = oracle.decoder(np.frombuffer(w, np.uint8))  # Init skips both atoms
        assert d.h and d.frame == 4096
        box = cont.build_stsd(ck, ch, 16, 44100)
        entry = 36 + 12 + len(ck)
        assert box[:16] == struct.pack(">I4sII", 16 + entry, b"stsd", 0, 1)
        assert box[16:52] == struct.pack(">I4s6xHIIHHHHI", entry, b"alac", 1, 0, 0, ch, 16, 0, 0, 44100 << 16)
        assert box[52:64] == struct.pack(">I4sI", 12 + len(ck), b"alac", 0) and box[64:] == ck
        assert cont.parse_stsd(box) == (ck, ch, 16, 44100 << 16)
        assert cont.parse_stsd(box[:40]) is None and cont.parse_stsd(b"x" * 100) is None
