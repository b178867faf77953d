"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and, when present, of the reference's
own compiled stage objects (oracle/_ref/libalacref.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libalacref.so")

i16p = C.POINTER(C.c_int16)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)

PC_FN = C.CFUNCTYPE(None, i32p, i32p, C.c_int32, i16p, C.c_int32, C.c_uint32, C.c_uint32)
COMP_FN = C.CFUNCTYPE(C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, i32p, u8p, u64p, C.c_int32,
                      C.c_int32, u32p)
DECOMP_FN = C.CFUNCTYPE(C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64, u64p, i32p,
                        C.c_int32, C.c_int32, u32p)


class Hooks(C.Structure):
    _fields_ = [("pc_block", PC_FN), ("unpc_block", PC_FN), ("dyn_comp", COMP_FN),
                ("dyn_decomp", DECOMP_FN)]


def build_oracle(force=False):
    """make -C oracle (liboracle.so always; _ref only where /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "alac_oracle.c")):
        subprocess.run(["make", "-C", ORACLE_DIR, "-s", "all"], check=True)
    return ORACLE_SO


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class Oracle:
    """The CPU restatement."""

    def __init__(self, path=None):
        self.lib = lib = C.CDLL(path or build_oracle())
        lib.oalac_pc_block.argtypes = lib.oalac_unpc_block.argtypes = \
            [i32p, i32p, C.c_int32, i16p, C.c_int32, C.c_uint32, C.c_uint32]
        lib.oalac_pc_block.restype = lib.oalac_unpc_block.restype = None
        lib.oalac_dyn_comp.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, i32p, u8p, u64p,
                                       C.c_int32, C.c_int32, u32p]
        lib.oalac_dyn_comp.restype = C.c_int32
        lib.oalac_dyn_decomp.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64, u64p,
                                         i32p, C.c_int32, C.c_int32, u32p]
        lib.oalac_dyn_decomp.restype = C.c_int32
        lib.oalac_mix.argtypes = [u8p, C.c_uint32, i32p, i32p, C.c_int32, C.c_int32, C.c_int32,
                                  u16p, C.c_int32]
        lib.oalac_unmix.argtypes = [i32p, i32p, u8p, C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                    u16p, C.c_int32]
        lib.oalac_put_bits.argtypes = [u8p, u64p, C.c_uint32, C.c_uint32]
        lib.oalac_get_bits.argtypes = [u8p, C.c_uint64, u64p, C.c_uint32]
        lib.oalac_get_bits.restype = C.c_uint32
        lib.oalac_init_coefs.argtypes = [i16p, C.c_uint32, C.c_int32]
        lib.oalac_encoder_new.argtypes = [C.c_uint32] * 4
        lib.oalac_encoder_new.restype = C.c_void_p
        lib.oalac_encoder_free.argtypes = [C.c_void_p]
        lib.oalac_encoder_set_hooks.argtypes = [C.c_void_p, C.POINTER(Hooks)]
        lib.oalac_encoder_set_fast_mode.argtypes = [C.c_void_p, C.c_int]
        lib.oalac_encoder_reset_state.argtypes = [C.c_void_p]
        lib.oalac_encoder_get_state.argtypes = [C.c_void_p, i16p]
        lib.oalac_encoder_set_state.argtypes = [C.c_void_p, i16p]
        lib.oalac_encode_packet.argtypes = [C.c_void_p, u8p, C.c_uint32, u8p, u32p]
        lib.oalac_encode_packet.restype = C.c_int32
        lib.oalac_max_packet_bytes.argtypes = [C.c_uint32] * 3
        lib.oalac_max_packet_bytes.restype = C.c_uint32
        lib.oalac_encoder_last_info.argtypes = [C.c_void_p, u32p]
        lib.oalac_magic_cookie.argtypes = [C.c_void_p, u8p]
        lib.oalac_magic_cookie.restype = C.c_uint32
        lib.oalac_magic_cookie_full.argtypes = [C.c_void_p, u8p]
        lib.oalac_magic_cookie_full.restype = C.c_uint32
        lib.oalac_channel_map.argtypes = [C.c_uint32]
        lib.oalac_channel_map.restype = C.c_uint32
        lib.oalac_encode_stream.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint32, u8p,
                                            C.c_uint64, u32p]
        lib.oalac_encode_stream.restype = C.c_int64
        lib.oalac_decoder_new.argtypes = [u8p, C.c_uint32, i32p]
        lib.oalac_decoder_new.restype = C.c_void_p
        lib.oalac_decoder_free.argtypes = [C.c_void_p]
        lib.oalac_decoder_set_hooks.argtypes = [C.c_void_p, C.POINTER(Hooks)]
        lib.oalac_decode_packet.argtypes = [C.c_void_p, u8p, C.c_uint32, u8p, u32p]
        lib.oalac_decode_packet.restype = C.c_int32
        lib.oalac_fnv1a64.argtypes = [u8p, C.c_uint64, C.c_uint64]
        lib.oalac_fnv1a64.restype = C.c_uint64

    # ---- stage level -------------------------------------------------------------------
    def pc_block(self, x, num, coefs, numactive, chanbits, denshift=9, fn=None):
        """returns (pc[len(x)], coefs_after); x is read up to max(num, numactive+1)."""
        x = np.ascontiguousarray(x, dtype=np.int32)
        pc = np.zeros(len(x) + 40, dtype=np.int32)
        xin = np.concatenate([x, np.zeros(40, np.int32)])
        co = np.array(coefs, dtype=np.int16).copy()
        (fn or self.lib.oalac_pc_block)(_ptr(xin, i32p), _ptr(pc, i32p), num, _ptr(co, i16p),
                                        numactive, chanbits, denshift)
        return pc[:len(x)], co

    def unpc_block(self, pc, num, coefs, numactive, chanbits, denshift=9, fn=None):
        pc = np.concatenate([np.ascontiguousarray(pc, dtype=np.int32), np.zeros(40, np.int32)])
        out = np.zeros(len(pc), dtype=np.int32)
        co = np.array(coefs, dtype=np.int16).copy()
        (fn or self.lib.oalac_unpc_block)(_ptr(pc, i32p), _ptr(out, i32p), num, _ptr(co, i16p),
                                          numactive, chanbits, denshift)
        return out[:len(pc) - 40], co

    def dyn_comp(self, pc, bit_size, start_bit=0, mb0=10, pb=40, kb=14, fn=None, fill=0):
        """returns (bytes covering the written bits, numBits)."""
        pc = np.ascontiguousarray(pc, dtype=np.int32)
        buf = np.full(len(pc) * 8 + 64, fill, dtype=np.uint8)
        pos = C.c_uint64(start_bit)
        nbits = C.c_uint32(0)
        st = (fn or self.lib.oalac_dyn_comp)(mb0, pb, kb, _ptr(pc, i32p), _ptr(buf, u8p),
                                             C.byref(pos), len(pc), bit_size, C.byref(nbits))
        assert st == 0, st
        assert pos.value == start_bit + nbits.value
        return buf[:(pos.value + 7) // 8].copy(), nbits.value

    def dyn_decomp(self, data, nbytes, num, max_size, start_bit=0, mb0=10, pb=40, kb=14, fn=None):
        buf = np.concatenate([np.ascontiguousarray(data, dtype=np.uint8), np.zeros(16, np.uint8)])
        pc = np.zeros(num + 8, dtype=np.int32)
        pos = C.c_uint64(start_bit)
        nbits = C.c_uint32(0)
        st = (fn or self.lib.oalac_dyn_decomp)(mb0, pb, kb, _ptr(buf, u8p), nbytes, C.byref(pos),
                                               _ptr(pc, i32p), num, max_size, C.byref(nbits))
        return st, pc[:num], nbits.value

    def mix(self, pcm, depth, n, mixbits, mixres, bytes_shifted):
        pcm = np.ascontiguousarray(pcm, dtype=np.uint8)
        u = np.zeros(n, np.int32)
        v = np.zeros(n, np.int32)
        sh = np.zeros(2 * n + 2, np.uint16)
        self.lib.oalac_mix(_ptr(pcm, u8p), depth, _ptr(u, i32p), _ptr(v, i32p), n, mixbits, mixres,
                           _ptr(sh, u16p), bytes_shifted)
        return u, v, sh[:2 * n]

    def unmix(self, u, v, depth, mixbits, mixres, sh, bytes_shifted):
        n = len(u)
        bps = {16: 2, 20: 3, 24: 3, 32: 4}[depth]
        out = np.zeros(n * 2 * bps, np.uint8)
        sh = np.ascontiguousarray(sh if sh is not None else np.zeros(2 * n, np.uint16), np.uint16)
        self.lib.oalac_unmix(_ptr(np.ascontiguousarray(u, np.int32), i32p),
                             _ptr(np.ascontiguousarray(v, np.int32), i32p), _ptr(out, u8p), depth,
                             n, mixbits, mixres, _ptr(sh, u16p), bytes_shifted)
        return out

    def fnv(self, data, seed=0):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        return int(self.lib.oalac_fnv1a64(_ptr(data, u8p), data.size, seed))

    # ---- drivers -----------------------------------------------------------------------
    def encoder(self, frame_size=4096, depth=16, channels=2, rate=44100, hooks=None, fast=False):
        """fast: SetFastMode(true) — stereo elements through EncodeStereoFast (no search)"""
        enc = OracleEncoder(self, frame_size, depth, channels, rate, hooks)
        if fast:
            self.lib.oalac_encoder_set_fast_mode(enc.h, 1)
        return enc

    def decoder(self, cookie, hooks=None):
        return OracleDecoder(self, cookie, hooks)


class OracleEncoder:
    def __init__(self, o, frame_size, depth, channels, rate, hooks=None):
        self.o, self.lib = o, o.lib
        self.frame_size, self.depth, self.channels = frame_size, depth, channels
        self.bpf = channels * {16: 2, 20: 3, 24: 3, 32: 4}[depth]
        self.h = self.lib.oalac_encoder_new(frame_size, depth, channels, rate)
        assert self.h
        self._hooks = hooks
        if hooks is not None:
            self.lib.oalac_encoder_set_hooks(self.h, C.byref(hooks))
        self.max_pkt = self.lib.oalac_max_packet_bytes(frame_size, depth, channels)
        self._out = np.zeros(self.max_pkt + 64, np.uint8)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oalac_encoder_free(self.h)
            self.h = None

    def reset(self):
        self.lib.oalac_encoder_reset_state(self.h)

    def get_state(self):
        s = np.zeros(64, np.int16)
        self.lib.oalac_encoder_get_state(self.h, _ptr(s, i16p))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, np.int16)
        self.lib.oalac_encoder_set_state(self.h, _ptr(s, i16p))

    def cookie(self):
        c = np.zeros(48, np.uint8)
        n = self.lib.oalac_magic_cookie_full(self.h, _ptr(c, u8p))
        return c[:n].copy()

    def encode_packet(self, pcm_bytes, num_samples):
        pcm = np.concatenate([np.ascontiguousarray(pcm_bytes, np.uint8).ravel(),
                              np.zeros(64 * self.bpf, np.uint8)])
        nb = C.c_uint32(0)
        st = self.lib.oalac_encode_packet(self.h, _ptr(pcm, u8p), num_samples, _ptr(self._out, u8p),
                                          C.byref(nb))
        assert st == 0, st
        return self._out[:nb.value].copy()

    def last_info(self):
        info = np.zeros(6, np.uint32)
        self.lib.oalac_encoder_last_info(self.h, _ptr(info, u32p))
        return dict(escape=int(info[0]), mixres=int(info[1]), numU=int(info[2]), numV=int(info[3]),
                    bitsU=int(info[4]), bitsV=int(info[5]))

    def encode_stream(self, pcm_bytes, total_samples, segment_packets=0):
        """returns (stream bytes, packet sizes)."""
        pcm = np.concatenate([np.ascontiguousarray(pcm_bytes, np.uint8).ravel(),
                              np.zeros(64 * self.bpf, np.uint8)])
        npk = (total_samples + self.frame_size - 1) // self.frame_size
        cap = npk * self.max_pkt + 64
        out = np.zeros(cap, np.uint8)
        sizes = np.zeros(max(npk, 1), np.uint32)
        n = self.lib.oalac_encode_stream(self.h, _ptr(pcm, u8p), total_samples, segment_packets,
                                         _ptr(out, u8p), cap, _ptr(sizes, u32p))
        assert n >= 0, n
        return out[:n].copy(), sizes[:npk].copy()


class OracleDecoder:
    def __init__(self, o, cookie, hooks=None):
        self.o, self.lib = o, o.lib
        ck = np.ascontiguousarray(cookie, np.uint8)
        st = C.c_int32(0)
        self.h = self.lib.oalac_decoder_new(_ptr(ck, u8p), ck.size, C.byref(st))
        self.status = st.value
        self._hooks = hooks
        if self.h and hooks is not None:
            self.lib.oalac_decoder_set_hooks(self.h, C.byref(hooks))
        cfg = bytes(ck)
        for atom in (b"frma", b"alac"):  # legacy wrappers, codec/ALACDecoder.cu:123-134
            if len(cfg) >= 12 and cfg[4:8] == atom:
                cfg = cfg[12:]
        self.frame = int.from_bytes(cfg[0:4], "big") if len(cfg) >= 24 else 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oalac_decoder_free(self.h)
            self.h = None

    def decode_packet(self, pkt, bytes_per_frame, frame_size=None):
        frame_size = frame_size or self.frame
        buf = np.concatenate([np.ascontiguousarray(pkt, np.uint8), np.zeros(16, np.uint8)])
        out = np.zeros(frame_size * bytes_per_frame + 64, np.uint8)
        ns = C.c_uint32(0)
        st = self.lib.oalac_decode_packet(self.h, _ptr(buf, u8p), len(pkt), _ptr(out, u8p),
                                          C.byref(ns))
        return st, out[:ns.value * bytes_per_frame].copy(), ns.value


class Ref:
    """The reference's own compiled C stage objects (only in the build container)."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = lib = C.CDLL(REF_SO)
        for f in (lib.pc_block, lib.unpc_block):
            f.argtypes = [i32p, i32p, C.c_int32, i16p, C.c_int32, C.c_uint32, C.c_uint32]
            f.restype = None
        lib.init_coefs.argtypes = [i16p, C.c_uint32, C.c_int32]
        lib.ref_dyn_comp_flat.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, i32p, u8p, u64p,
                                          C.c_int32, C.c_int32, u32p]
        lib.ref_dyn_comp_flat.restype = C.c_int32
        lib.ref_dyn_decomp_flat.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint64,
                                            u64p, i32p, C.c_int32, C.c_int32, u32p]
        lib.ref_dyn_decomp_flat.restype = C.c_int32
        lib.ref_put_bits.argtypes = [u8p, u64p, C.c_uint32, C.c_uint32]
        lib.ref_get_bits16.argtypes = [u8p, u64p, C.c_uint32]
        lib.ref_get_bits16.restype = C.c_uint32

    def hooks(self):
        lib = self.lib
        return Hooks(C.cast(lib.pc_block, PC_FN), C.cast(lib.unpc_block, PC_FN),
                     C.cast(lib.ref_dyn_comp_flat, COMP_FN),
                     C.cast(lib.ref_dyn_decomp_flat, DECOMP_FN))


def have_ref():
    return os.path.exists(REF_SO)


def read_wav(path):
    """minimal RIFF/WAVE PCM reader -> (channels, rate, bits, data bytes as uint8 array)."""
    with open(path, "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos = 12
    fmt = None
    while pos + 8 <= len(b):
        cid = b[pos:pos + 4]
        sz = int.from_bytes(b[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            ch = int.from_bytes(b[pos + 10:pos + 12], "little")
            rate = int.from_bytes(b[pos + 12:pos + 16], "little")
            bits = int.from_bytes(b[pos + 22:pos + 24], "little")
            fmt = (ch, rate, bits)
        elif cid == b"data":
            data = np.frombuffer(b, np.uint8, min(sz, len(b) - pos - 8), pos + 8)
            return fmt + (data,)
        pos += 8 + sz + (sz & 1)
    raise ValueError("no data chunk")


# ---- > 2 channels: test data and the element view of a packet ------------------------------------

def channel_elements(o, channels):
    """[(first channel, channels in the element)] in packet order, from sChannelMaps (codec/ALACEncoder.cu:97-107)."""
    m, out, ci = o.lib.oalac_channel_map(channels), [], 0
    while ci < channels:
        n = 2 if ((m >> (3 * ci)) & 7) == 1 else 1
        out.append((ci, n))
        ci += n
    return out


def interleave_channels(parts, depth):
    """parts: list of (pcm bytes, channels) with the same number of frames -> one interleaved byte array."""
    bps = {16: 2, 20: 3, 24: 3, 32: 4}[depth]
    cols = [np.ascontiguousarray(p, np.uint8).reshape(-1, c * bps) for p, c in parts]
    return np.concatenate(cols, axis=1).reshape(-1)


def take_channels(pcm, channels, first, count, depth):
    bps = {16: 2, 20: 3, 24: 3, 32: 4}[depth]
    a = np.ascontiguousarray(pcm, np.uint8).reshape(-1, channels * bps)
    return np.ascontiguousarray(a[:, first * bps:(first + count) * bps]).reshape(-1)


def splice_elements(packets):
    """packets: list of (complete one-element packet bytes, instance tag).  Returns the packet that carries all the
    elements: every element's bits up to its ID_END (the last set bit run '111' before the zero padding), the 4-bit
    instance tag replaced, then one ID_END and the byte alignment (codec/ALACEncoder.cu:1034-1039)."""
    bits = []
    for pk, tag in packets:
        b = np.unpackbits(np.ascontiguousarray(pk, np.uint8))
        last = int(np.nonzero(b)[0][-1])  # last bit of ID_END
        e = b[:last - 2].copy()
        e[3:7] = [(tag >> 3) & 1, (tag >> 2) & 1, (tag >> 1) & 1, tag & 1]
        bits.append(e)
    bits.append(np.array([1, 1, 1], np.uint8))
    return np.packbits(np.concatenate(bits))
