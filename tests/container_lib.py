"""ctypes binding of convert-utility/libcontainer.so (the product's container code, CPU only) + synthetic PCM."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CU = os.path.join(ROOT, "convert-utility")
SO = os.path.join(CU, "libcontainer.so")


class Info(C.Structure):
    _fields_ = [("kind", C.c_int32), ("is_alac", C.c_int32), ("big_endian_pcm", C.c_int32), ("sample_rate", C.c_double),
                ("channels", C.c_uint32), ("bits_per_channel", C.c_uint32), ("alac_source_flag", C.c_uint32),
                ("frames_per_packet", C.c_uint32), ("data_pos", C.c_uint64), ("data_size", C.c_uint64)]


def _u8(b):
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")


class Container:
    def __init__(self):
        src = [os.path.join(CU, n) for n in ("container.cpp", "container.h", "container_capi.cpp")]
        if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src):
            subprocess.check_call(["make", "-C", CU, "libcontainer.so"], stdout=subprocess.DEVNULL)
        self.lib = lib = C.CDLL(SO)
        lib.alacfile_sniff.restype = C.c_int32
        lib.alacfile_build_alac_caf.restype = C.c_uint64
        lib.alacfile_build_alac_caf.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p,
                                                C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                                C.c_uint64]
        for f in (lib.alacfile_build_wave, lib.alacfile_build_pcm_caf):
            f.restype = C.c_uint64
            f.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        lib.alacfile_parse_alac_caf.restype = C.c_int64
        lib.alacfile_parse_alac_caf.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                C.c_void_p]
        lib.alacfile_append_ber.restype = C.c_uint32
        lib.alacfile_read_ber.restype = C.c_uint32

    def sniff(self, data):
        info, err = Info(), C.create_string_buffer(128)
        rc = self.lib.alacfile_sniff(_u8(data), C.c_uint64(len(data)), C.byref(info), err, 128)
        return rc, info, err.value.decode()

    def build_alac_caf(self, rate, ch, bits, input_bytes, cookie, sizes, stream):
        sz = np.ascontiguousarray(sizes, np.uint32)
        cap = len(stream) + 4096 + 8 * len(sz) + 3 * (input_bytes // (4096 * ch * (bits >> 3)) + 2)
        out = (C.c_uint8 * cap)()
        n = self.lib.alacfile_build_alac_caf(rate, ch, bits, 4096, input_bytes, _u8(cookie), len(cookie),
                                             sz.ctypes.data if len(sz) else None, len(sz), _u8(stream), len(stream), out, cap)
        assert n <= cap
        return bytes(out[:n])

    def build_wave(self, rate, ch, bits, pcm, caf=False):
        cap = len(pcm) + 256
        out = (C.c_uint8 * cap)()
        fn = self.lib.alacfile_build_pcm_caf if caf else self.lib.alacfile_build_wave
        n = fn(rate, ch, bits, _u8(pcm), len(pcm), out, cap)
        return bytes(out[:n])

    def parse_alac_caf(self, data, max_packets=1 << 16):
        cookie = (C.c_uint8 * 64)()
        csize, dpos = C.c_uint32(0), C.c_uint64(0)
        sizes = np.zeros(max_packets, np.uint32)
        n = self.lib.alacfile_parse_alac_caf(_u8(data), len(data), cookie, C.byref(csize), sizes.ctypes.data, max_packets,
                                             C.byref(dpos))
        if n < 0:
            return None
        return bytes(cookie[:csize.value]), sizes[:n].copy(), dpos.value

    def cookie(self, kind, cookie):
        out = (C.c_uint8 * 128)()
        self.lib.alacfile_cookie.restype = C.c_uint64
        n = self.lib.alacfile_cookie(kind, _u8(cookie), len(cookie), out, C.c_uint64(128))
        return bytes(out[:n])

    def build_stsd(self, cookie, ch, bits, rate):
        out = (C.c_uint8 * 256)()
        self.lib.alacfile_build_stsd.restype = C.c_uint64
        n = self.lib.alacfile_build_stsd(_u8(cookie), len(cookie), ch, bits, rate, out, C.c_uint64(256))
        return bytes(out[:n])

    def parse_stsd(self, box):
        ck, f = (C.c_uint8 * 64)(), (C.c_uint32 * 3)()
        self.lib.alacfile_parse_stsd.restype = C.c_int64
        n = self.lib.alacfile_parse_stsd(_u8(box), C.c_uint64(len(box)), ck, f)
        return None if n < 0 else (bytes(ck[:n]), f[0], f[1], f[2])

    def build_alac_m4a(self, rate, ch, bits, total_frames, cookie, sizes, stream, frames_per_packet=4096):
        sz = np.ascontiguousarray(sizes, np.uint32)
        cap = len(stream) + 4096 + 4 * len(sz)
        out = (C.c_uint8 * cap)()
        self.lib.alacfile_build_alac_m4a.restype = C.c_uint64
        self.lib.alacfile_build_alac_m4a.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p,
                                                     C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                                     C.c_uint64]
        n = self.lib.alacfile_build_alac_m4a(rate, ch, bits, frames_per_packet, total_frames, _u8(cookie), len(cookie),
                                             sz.ctypes.data if len(sz) else None, len(sz), _u8(stream), len(stream), out, cap)
        assert n <= cap
        return bytes(out[:n])

    def parse_alac_m4a(self, data, max_packets=1 << 16):
        """-> (info, cookie, sizes, positions) or the diagnostic string"""
        info, err = Info(), C.create_string_buffer(160)
        cookie, csize = (C.c_uint8 * 64)(), C.c_uint32(0)
        sizes, pos = np.zeros(max_packets, np.uint32), np.zeros(max_packets, np.uint64)
        self.lib.alacfile_parse_alac_m4a.restype = C.c_int64
        self.lib.alacfile_parse_alac_m4a.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        n = self.lib.alacfile_parse_alac_m4a(_u8(data), len(data), C.byref(info), cookie, C.byref(csize), sizes.ctypes.data,
                                             pos.ctypes.data, max_packets, err, 160)
        if n < 0:
            return err.value.decode()
        return info, bytes(cookie[:csize.value]), sizes[:n].copy(), pos[:n].copy()

    def ber(self, v):
        out = (C.c_uint8 * 5)()
        n = self.lib.alacfile_append_ber(C.c_uint32(v), out)
        return bytes(out[:n])

    def read_ber(self, b):
        used = C.c_uint32(0)
        v = self.lib.alacfile_read_ber(_u8(b), len(b), C.byref(used))
        return v, used.value


def music_like(n_frames, channels, bits, seed):
    """deterministic band-limited noise + tones, full scale / 4, packed little endian"""
    rng = np.random.default_rng(seed)
    t = np.arange(n_frames)
    out = []
    for c in range(channels):
        x = 0.2 * np.sin(2 * np.pi * (220.0 + 3 * seed + 40 * c) * t / 44100.0) + 0.1 * np.sin(2 * np.pi * 1310.0 * t / 44100.0)
        noise = rng.standard_normal(max(n_frames, 8))
        noise = np.convolve(noise, np.ones(8) / 8, mode="same")[:n_frames] * 0.05
        out.append(x + noise)
    a = np.stack(out, axis=1)
    full = float(1 << (bits - 1)) - 1
    v = np.round(a * full).astype(np.int64)
    if bits == 16:
        return v.astype("<i2").tobytes()
    if bits == 32:
        return v.astype("<i4").tobytes()
    b = (v & 0xffffff).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3]
    return b.tobytes()
