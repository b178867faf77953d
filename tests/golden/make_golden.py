#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REFERENCE's own compiled stage objects
(oracle/_ref/libalacref.so, built by oracle/Makefile from /root/reference/codec/*.c).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py

Outputs (data only — inputs and expected outputs, no reference source):
  stage_vectors.npz   pc_block / dyn_comp inputs and the reference's outputs for every call shape of
                      the hot path
  packets.npz         short excerpts of the reference's audio/50.wav (stereo) and audio/05.wav (mono)
                      with the packets produced by the encoder driver running over the reference's
                      pc_block/dyn_comp, chained and independent
  known_answers.json  sizes / FNV-1a-64 of whole-file encodes of the three reference WAVs and of the
                      synthetic workload (pins the generator too)
  forged.npz          packets with FOREIGN header / cookie parameters (oracle/forge.py run over the reference's compiled
                      pc_block / dyn_comp / BitBufferWrite) and the PCM the decoder driver produces from them over the
                      reference's dyn_decomp / unpc_block: the pin of the decoder's general paths
  caf_headers.json    chunk bytes, BER codes and base packet tables from the reference's own CAFFileALAC.cpp
  wav50_pcm.xz,       the sample data of the reference's audio/50.wav (stereo, 237 packets) and audio/05.wav (mono, 302
  wav05_pcm.xz        packets), xz-compressed: the INPUTS of the whole-file known answers, so that the GPU box (which has
                      no /root/reference) can encode the full-length chains (tests/test_gpu_wholefile.py)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle_lib import Oracle, Ref, read_wav  # noqa: E402

REF_AUDIO = "/root/reference/audio"


def signal(rng, kind, n, chanbits):
    lim = (1 << (chanbits - 1)) - 1
    if kind == 0:
        x = rng.integers(-lim, lim + 1, size=n)
    elif kind == 1:
        x = np.cumsum(rng.integers(-40, 41, size=n))
    elif kind == 2:
        t = np.arange(n)
        x = (0.4 * lim * np.sin(t * 0.05) + rng.integers(-20, 21, size=n)).astype(np.int64)
    elif kind == 3:
        x = np.zeros(n, np.int64)
        if n:
            x[rng.integers(0, n, size=max(n // 60, 1))] = rng.integers(-lim, lim + 1, size=max(n // 60, 1))
    else:
        x = rng.integers(-3, 4, size=n)
    return np.clip(x, -lim - 1, lim).astype(np.int32)


def make_stage_vectors(o, r):
    rng = np.random.default_rng(20261004)
    out = {}
    meta = []
    # (num, numactive) shapes of the hot path + the general path ("deep LPC", SURVEY §7 hard part 6)
    shapes = [(512, 8), (128, 4), (128, 8), (4096, 4), (4096, 8), (3, 8), (12, 8), (0, 4)]
    shapes += [(600, na) for na in (1, 2, 3, 5, 16, 30, 31, 0)]
    i = 0
    for num, na in shapes:
        for chanbits in (16, 17, 20, 21):
            if num == 4096 and chanbits in (20,):
                continue
            kind = i % 5
            x = signal(rng, kind, max(num, 40) + 8, chanbits)
            co = np.zeros(32, np.int16)
            if i % 3 == 0:
                r.lib.init_coefs(co.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_int16)), 9, 16)
            else:
                co[:] = rng.integers(-3000, 3000, size=32)
            if i % 11 == 0:
                co[:8] = 32767 - rng.integers(0, 3, size=8)  # exercises the int16 wrap
            pc, cafter = o.pc_block(x, num, co, na, chanbits, fn=r.lib.pc_block)
            out[f"pc{i}_x"] = x
            out[f"pc{i}_coefs"] = co
            out[f"pc{i}_pc"] = pc
            out[f"pc{i}_after"] = cafter
            meta.append(dict(kind="pc", id=i, num=num, numactive=na, chanbits=chanbits))
            i += 1
    j = 0
    for n in (0, 1, 7, 128, 512, 4096):
        for bits in (16, 17, 20, 21, 24, 32):
            if n == 4096 and bits not in (16, 17, 21):
                continue
            for kind in ((j % 5), 3):
                pc = signal(rng, kind, n, bits)
                if kind == 3 and n >= 512:
                    pc[n // 2:n // 2 + 8] = (1 << (bits - 1)) - 1  # n > 0xffff clamp for wide samples
                sb = int(rng.integers(0, 8))
                data, nbits = o.dyn_comp(pc, bits, start_bit=sb, fn=r.lib.ref_dyn_comp_flat)
                out[f"ag{j}_pc"] = pc
                out[f"ag{j}_bytes"] = data
                meta.append(dict(kind="ag", id=j, n=n, bits=bits, start_bit=sb, nbits=nbits))
                j += 1
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "stage_vectors.npz"), **out)
    print("stage_vectors.npz:", i, "pc cases,", j, "ag cases")


def encode_all(o, hooks, depth, ch, rate, data, total, seg):
    e = o.encoder(4096, depth, ch, rate, hooks)
    return e.encode_stream(data, total, seg)


def make_packets(o, r):
    H = r.hooks()
    out = {}
    # stereo excerpt: 8 packets from a loud region of 50.wav, + a 1000-sample partial tail
    ch, rate, bits, data = read_wav(os.path.join(REF_AUDIO, "50.wav"))
    first, npk, tail = 100, 8, 1000
    seg = data[first * 16384:(first + npk) * 16384 + tail * 4].copy()
    total = npk * 4096 + tail
    out["stereo_pcm"] = seg
    s, sz = encode_all(o, H, bits, ch, rate, seg, total, 0)
    out["stereo_chained_stream"], out["stereo_chained_sizes"] = s, sz
    s, sz = encode_all(o, H, bits, ch, rate, seg, total, 1)
    out["stereo_indep_stream"], out["stereo_indep_sizes"] = s, sz
    # mono excerpt from 05.wav
    ch, rate, bits, data = read_wav(os.path.join(REF_AUDIO, "05.wav"))
    first, npk, tail = 120, 8, 1904
    seg = data[first * 8192:(first + npk) * 8192 + tail * 2].copy()
    total = npk * 4096 + tail
    out["mono_pcm"] = seg
    s, sz = encode_all(o, H, bits, ch, rate, seg, total, 0)
    out["mono_chained_stream"], out["mono_chained_sizes"] = s, sz
    s, sz = encode_all(o, H, bits, ch, rate, seg, total, 1)
    out["mono_indep_stream"], out["mono_indep_sizes"] = s, sz
    np.savez_compressed(os.path.join(HERE, "packets.npz"), **out)
    print("packets.npz written")


FORGED_STREAMS = [  # (depth, channels, frame size, cookie pb, mb, kb, packets)
    (16, 2, 256, 40, 10, 14, 40), (16, 2, 256, 20, 5, 9, 24), (16, 1, 256, 63, 30, 16, 24), (24, 2, 128, 40, 10, 14, 24),
    (24, 1, 128, 30, 12, 11, 16), (20, 2, 128, 40, 10, 14, 16), (20, 1, 96, 50, 8, 13, 12), (32, 2, 64, 40, 10, 14, 16),
    (32, 1, 64, 25, 10, 12, 12),
]


def make_forged(o, r):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import forge
    fr = forge.Forger(o, dict(pc_block=r.lib.pc_block, dyn_comp=r.lib.ref_dyn_comp_flat, put_bits=r.lib.ref_put_bits))
    out, meta = {}, []
    for si, (depth, ch, frame, pb, mb, kb, count) in enumerate(FORGED_STREAMS):
        rng = np.random.default_rng(4000 + si)
        pk, pcm, ok = forge.forge_batch(fr, rng, count, depth, ch, frame, pb, mb, kb)
        ck = forge.cookie(frame, depth, ch, pb, mb, kb)
        dec = o.decoder(ck, hooks=r.hooks())
        bpf = ch * forge.BPS[depth]
        want = []
        for a, p, k in zip(pk, pcm, ok):
            st, w, n = dec.decode_packet(a, bpf)
            assert st == 0 and n * bpf == len(p)
            assert (not k) or np.array_equal(w, p)
            want.append(w)
        out[f"s{si}_stream"] = np.concatenate(pk)
        out[f"s{si}_sizes"] = np.array([len(a) for a in pk], np.uint32)
        out[f"s{si}_pcm"] = np.concatenate(want)
        out[f"s{si}_cookie"] = ck
        meta.append(dict(id=si, depth=depth, channels=ch, frame=frame, pb=pb, mb=mb, kb=kb, packets=count,
                         lossless=[bool(k) for k in ok]))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "forged.npz"), **out)
    print("forged.npz:", sum(m["packets"] for m in meta), "packets,", os.path.getsize(os.path.join(HERE, "forged.npz")), "bytes")


def make_wav_pcm_fixtures():
    import lzma
    for name, out in (("50.wav", "wav50_pcm.xz"), ("05.wav", "wav05_pcm.xz")):
        ch, rate, bits, data = read_wav(os.path.join(REF_AUDIO, name))
        with open(os.path.join(HERE, out), "wb") as f:
            f.write(lzma.compress(data.tobytes(), preset=9 | lzma.PRESET_EXTREME))
        print(out, "written:", os.path.getsize(os.path.join(HERE, out)), "bytes for", data.size, "bytes of PCM")


def make_caf_headers():
    """CAF chunk bytes / BER codes / base packet tables from the reference's own CAFFileALAC.cpp (oracle/_ref/libcafref.so)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import test_container_refpin as t
    with open(os.path.join(HERE, "caf_headers.json"), "w") as f:
        json.dump(t.collect(t.CafRef()), f, indent=0, sort_keys=True)
    print("caf_headers.json written")


def make_known_answers(o, r):
    import alac_amd
    H = r.hooks()
    ka = {"fnv": "FNV-1a 64-bit over the concatenated packets", "wav": {}, "synthetic": {}}
    for name in ("50.wav", "70.wav", "05.wav"):
        ch, rate, bits, data = read_wav(os.path.join(REF_AUDIO, name))
        total = len(data) // (ch * bits // 8)
        s, sz = encode_all(o, H, bits, ch, rate, data, total, 0)
        s1, sz1 = encode_all(o, H, bits, ch, rate, data, total, 1)
        ka["wav"][name] = dict(channels=ch, bits=bits, rate=rate, sample_frames=total, packets=len(sz),
                               chained_bytes=int(len(s)), chained_fnv=f"{o.fnv(s):016x}",
                               chained_first_sizes=[int(x) for x in sz[:8]], chained_last_size=int(sz[-1]),
                               indep_bytes=int(len(s1)), indep_fnv=f"{o.fnv(s1):016x}")
    for depth, ch in ((16, 2), (24, 2), (20, 2), (32, 2), (16, 1), (24, 1)):
        fmt = alac_amd.make_format(4096, depth, ch)
        n = 64
        pcm = alac_amd.synth_pcm(0, n, fmt)
        s, sz = encode_all(o, H, depth, ch, 44100, pcm, n * 4096, 1)
        ka["synthetic"][f"{depth}bit_{ch}ch"] = dict(frames=n, pcm_fnv=f"{o.fnv(pcm):016x}", bytes=int(len(s)),
                                                     fnv=f"{o.fnv(s):016x}", sizes=[int(x) for x in sz])
    # the all-zero packet: 32 bytes stereo, 19 bytes mono
    e = o.encoder(4096, 16, 2, 44100, H)
    ka["silent_stereo_packet"] = e.encode_packet(np.zeros(16384, np.uint8), 4096).tobytes().hex()
    e = o.encoder(4096, 16, 1, 44100, H)
    ka["silent_mono_packet"] = e.encode_packet(np.zeros(8192, np.uint8), 4096).tobytes().hex()
    e = o.encoder(4096, 16, 2, 44100)
    ka["cookie_16bit_stereo_44k1"] = e.cookie().tobytes().hex()
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("known_answers.json written")


if __name__ == "__main__":
    o, r = Oracle(), Ref()
    make_stage_vectors(o, r)
    make_packets(o, r)
    make_known_answers(o, r)
    make_forged(o, r)
    make_wav_pcm_fixtures()
    make_caf_headers()
