"""ctypes binding of libalac_hip.so (the C-ABI declared in include/alac_hip.h).

PyTorch is used only as the owner of device memory and streams: every call below hands raw device
pointers to the HIP library.  There is no CPU fallback: if the library or a GPU is missing the
import of the library / creation of a context raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALAC_HIP_LIB") or os.path.join(_HERE, "libalac_hip.so")  # ALAC_HIP_LIB: A/B runs against another build
SYNTH_PATH = os.path.join(_HERE, "libalac_synth.so")

STATE_INT16 = 64
BPS = {16: 2, 20: 3, 24: 3, 32: 4}


class Format(C.Structure):
    _fields_ = [("frame_size", C.c_uint32), ("bit_depth", C.c_uint32), ("num_channels", C.c_uint32),
                ("sample_rate", C.c_uint32)]

    @property
    def bytes_per_frame(self):
        return self.num_channels * BPS[self.bit_depth]

    @property
    def packet_bytes(self):
        return self.frame_size * self.bytes_per_frame


def make_format(frame_size=4096, bit_depth=16, num_channels=2, sample_rate=44100):
    return Format(frame_size, bit_depth, num_channels, sample_rate)


_vp, _u32, _i32, _u64 = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64

# name -> (restype, argtypes); mirrors include/alac_hip.h one to one
SIGNATURES = {
    "alac_hip_device_count": (_i32, []),
    "alac_hip_create": (_i32, [C.POINTER(_vp), _i32, _vp]),
    "alac_hip_destroy": (None, [_vp]),
    "alac_hip_synchronize": (_i32, [_vp]),
    "alac_hip_last_error": (C.c_char_p, [_vp]),
    "alac_hip_encode_regime": (C.c_char_p, [_vp, C.POINTER(Format), _u32]),
    "alac_hip_debug_waves_offset": (_u64, [C.POINTER(Format), _u32, _u32]),
    "alac_hip_set_option": (_i32, [_vp, C.c_char_p, _i32]),
    "alac_hip_get_option": (_i32, [_vp, C.c_char_p, C.POINTER(_i32)]),
    "alac_hip_stream": (_vp, [_vp]),
    "alac_hip_encode_workspace_bytes": (_u64, [C.POINTER(Format), _u32, _u32]),
    "alac_hip_encode_max_output_bytes": (_u64, [C.POINTER(Format), _u32]),
    "alac_hip_encode": (_i32, [_vp, C.POINTER(Format), _vp, _vp, _u32, _vp, _u32, _vp, _i32, _vp, _u64,
                               _vp, _u64, _vp, _vp]),
    "alac_hip_encode_segmented": (_i32, [_vp, C.POINTER(Format), _vp, _vp, _u32, _vp, _u32, _u32, _vp, _i32, _vp, _u64,
                                         _vp, _u64, _vp, _vp]),
    "alac_hip_profile_begin": (_i32, [_vp, _u32]),
    "alac_hip_profile_end": (_i32, [_vp, C.POINTER(_u32), C.POINTER(C.c_float), C.POINTER(_u32)]),
    "alac_hip_num_stages": (_u32, []),
    "alac_hip_stage_name": (C.c_char_p, [_u32]),
    "alac_hip_magic_cookie": (_u32, [C.POINTER(Format), _u32, _u32, _vp]),
    "alac_hip_magic_cookie_size": (_u32, [C.POINTER(Format)]),
    "alac_hip_magic_cookie_full": (_u32, [C.POINTER(Format), _u32, _u32, _vp, _u32]),
    "alac_hip_state_int16": (_u32, [C.POINTER(Format)]),
    "alac_hip_decode_workspace_bytes": (_u64, [C.POINTER(Format), _u32]),
    "alac_hip_decode_workspace_bytes_stream": (_u64, [C.POINTER(Format), _u32, _u64]),
    "alac_hip_decode": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, _vp, _vp]),
    "alac_hip_format_from_cookie": (_i32, [_vp, _u32, C.POINTER(Format)]),
    "alac_hip_pc_block": (_i32, [_vp, _vp, _vp, _u32, _u32, _i32, _vp, _i32, _u32, _u32]),
    "alac_hip_unpc_block": (_i32, [_vp, _vp, _vp, _u32, _u32, _i32, _vp, _i32, _u32, _u32]),
    "alac_hip_dyn_comp": (_i32, [_vp, _u32, _u32, _u32, _vp, _u32, _u32, _i32, _i32, _vp, _u32, _vp]),
    "alac_hip_dyn_decomp": (_i32, [_vp, _u32, _u32, _u32, _vp, _u32, _u32, _vp, _u32, _i32, _i32, _vp, _vp]),
    "alac_hip_encode_host": (_i32, [_vp, C.POINTER(Format), _vp, _u64, _u32, _vp, _i32, _vp, _u64, _vp,
                                    C.POINTER(_u64)]),
    "alac_hip_encode_host_segments": (_i32, [_vp, C.POINTER(Format), _vp, _vp, _u32, _vp, _u32, _vp, _i32, _vp, _u64, _vp,
                                             C.POINTER(_u64)]),
    "alac_hip_decode_host": (_i32, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _vp, _vp]),
    "alac_synth_frame": (None, [_u64, _u32, _u32, _u32, _vp]),
    "alac_synth_pcm": (None, [_u64, _u32, _u32, _u32, _u32, _vp]),
    "alac_hip_synth_pcm": (_i32, [_vp, _u64, _u32, C.POINTER(Format), _vp]),
    "alac_hip_shard_range": (_i32, [_u64, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "alac_hip_shard_offsets": (_i32, [_vp, _u32, _vp]),
    "alac_hip_comm_unique_id": (_i32, [_vp]),
    "alac_hip_comm_create": (_i32, [C.POINTER(_vp), _i32, _vp, _u32, _u32]),
    "alac_hip_comm_destroy": (None, [_vp]),
    "alac_hip_comm_rank": (_u32, [_vp]),
    "alac_hip_comm_world": (_u32, [_vp]),
    "alac_hip_comm_last_error": (C.c_char_p, [_vp]),
    "alac_hip_reassemble_begin": (_i32, [_vp, _u32, _vp, _u64, _u64, _vp, _u32, _vp, _vp]),
    "alac_hip_reassemble_finish": (_i32, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "alac_hip_reassemble": (_i32, [_vp, _vp, _vp, _u64, _vp, _u32, _vp, _vp, _u64, _vp, _vp]),
}

COMM_ID_BYTES = 128
COMM_SLOTS = 4

_lib = None


def load_library(path=None):
    """dlopen libalac_hip.so and bind every symbol of include/alac_hip.h (raises if one is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} is missing: build it with `make -C alac_amd/csrc` (or __graft_entry__.build()); "
            "the HIP path has no CPU fallback")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def source_fingerprint():
    """sha256 (first 16 hex digits) over the kernel and C-ABI sources under alac_amd/csrc + include/: what a set of PMC
    counters under profiles/ was collected on (bench.py refuses counters of other sources; no GPU or library needed)"""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = []
    for d in (os.path.join(_HERE, "csrc"), os.path.join(root, "include")):
        for dp, _, fn in os.walk(d):
            files += [os.path.join(dp, f) for f in fn if f.endswith((".hip", ".hpp", ".h", ".cpp", ".c"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def shard_range(num_units, world, rank):
    """(first, count) of the units rank `rank` of `world` encodes (alac_hip_shard_range); raises on bad arguments"""
    first, count = _u64(), _u64()
    if load_library().alac_hip_shard_range(num_units, world, rank, C.byref(first), C.byref(count)) != 0:
        raise ValueError(f"bad shard arguments: world {world}, rank {rank}")
    return first.value, count.value


def shard_offsets(shard_bytes):
    """byte offsets of the shards in the re-assembled stream, [world + 1] (alac_hip_shard_offsets)"""
    n = len(shard_bytes)
    src = (_u64 * n)(*[int(x) for x in shard_bytes])
    dst = (_u64 * (n + 1))()
    if load_library().alac_hip_shard_offsets(src, n, dst) != 0:
        raise ValueError("bad shard sizes")
    return list(dst)


def synth_pcm(first_frame, num_frames, fmt):
    """Deterministic synthetic PCM (alac_synth.c) as a uint8 numpy array, host side, no GPU needed."""
    p = SYNTH_PATH if os.path.exists(SYNTH_PATH) else LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{SYNTH_PATH} missing: run `make -C alac_amd/csrc`")
    lib = C.CDLL(p)
    lib.alac_synth_pcm.argtypes = SIGNATURES["alac_synth_pcm"][1]
    lib.alac_synth_pcm.restype = None
    out = np.zeros(num_frames * fmt.packet_bytes, dtype=np.uint8)
    lib.alac_synth_pcm(first_frame, num_frames, fmt.frame_size, fmt.bit_depth, fmt.num_channels,
                       out.ctypes.data)
    return out


class AlacError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"alac_hip status {code}: {msg}")
        self.code = code


class Context:
    """One alac_hip_ctx bound to a device, enqueuing on a torch stream of its own (`self.stream`).

    Stream discipline: the library's kernels run on `self.stream`; every call below first makes that stream wait for the
    caller's current torch stream (inputs produced there — fills, copies — are complete) and afterwards makes the caller's
    stream wait for it (events the caller records on its stream, e.g. bench.py's hand-over to the RCCL side stream, are
    ordered behind the call), and tensors allocated inside a call are allocated and filled on `self.stream`.  Without this
    the library would sit on a private non-blocking stream that nothing the caller does is ordered against."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("alac_amd needs a HIP device: no GPU visible and there is no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        h = _vp()
        rc = self.lib.alac_hip_create(C.byref(h), device, _vp(self.stream.cuda_stream))
        if rc != 0:
            raise AlacError(rc, "alac_hip_create failed")
        self.h = h
        self._ws = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.alac_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self):
        """context manager around one library call: see the class docstring"""
        import contextlib
        t = self.torch

        @contextlib.contextmanager
        def cm():
            cur = t.cuda.current_stream(self.device)
            if cur == self.stream:  # the caller already works on the context's stream (bench.py's timed loop): nothing to order
                yield cur
                return
            self.stream.wait_stream(cur)
            with t.cuda.stream(self.stream):
                yield cur
            cur.wait_stream(self.stream)
        return cm()

    def _check(self, rc):
        if rc != 0:
            raise AlacError(rc, self.lib.alac_hip_last_error(self.h).decode())

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
        return self._ws

    def synchronize(self):
        self._check(self.lib.alac_hip_synchronize(self.h))

    def regime(self, fmt, num_segments):
        """alac_hip_encode_regime: "throughput" / "latency" / "tiny" / "lane" for a batch of independent segments"""
        return self.lib.alac_hip_encode_regime(self.h, C.byref(fmt), int(num_segments)).decode()

    def set_option(self, key, value):
        """alac_hip_set_option: pin a code path of this context (include/alac_hip.h lists the keys)"""
        self._check(self.lib.alac_hip_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = _i32(0)
        self._check(self.lib.alac_hip_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """context manager: set options for the duration of a with-block, then restore them"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return cm()

    def synth_pcm(self, first_frame, num_frames, fmt, out=None):
        """The synthetic PCM of frames [first_frame, first_frame + num_frames) generated ON THE DEVICE (same bytes as
        synth_pcm(): one generator source) -> uint8 cuda tensor."""
        with self._call() as cur:
            t = self.torch
            if out is None:
                out = t.empty(num_frames * fmt.packet_bytes, dtype=t.uint8, device=self.device)
            assert out.numel() >= num_frames * fmt.packet_bytes
            self._check(self.lib.alac_hip_synth_pcm(self.h, first_frame, num_frames, C.byref(fmt), out.data_ptr()))
            out.record_stream(cur)
            return out

    # ---- encode ----------------------------------------------------------------------------
    def encode_buffers(self, fmt, num_packets):
        """Preallocate outputs so a timed loop does no allocation."""
        t = self.torch
        cap = int(self.lib.alac_hip_encode_max_output_bytes(C.byref(fmt), num_packets))
        return dict(out=t.empty(cap, dtype=t.uint8, device=self.device),
                    sizes=t.empty(num_packets, dtype=t.int32, device=self.device),
                    offsets=t.empty(num_packets + 1, dtype=t.int64, device=self.device))

    def encode(self, fmt, pcm, num_packets, num_samples=None, seg_first=None, state=None, state_in=False,
               bufs=None, max_segment_packets=0):
        """pcm: uint8 cuda tensor (num_packets * fmt.packet_bytes).  Returns the buffers dict
        (out, sizes, offsets); offsets[-1] is the total byte count.  Asynchronous — with a segment table only if
        max_segment_packets (the caller's bound on the longest segment, alac_hip_encode_segmented) is given."""
        with self._call() as cur:
            t = self.torch
            assert pcm.is_cuda and pcm.dtype == t.uint8 and pcm.numel() >= num_packets * fmt.packet_bytes
            nseg = num_packets if seg_first is None else seg_first.numel() - 1
            bufs = bufs or self.encode_buffers(fmt, num_packets)
            wsb = int(self.lib.alac_hip_encode_workspace_bytes(C.byref(fmt), num_packets, nseg))
            ws = self._workspace(wsb)
            rc = self.lib.alac_hip_encode_segmented(
                self.h, C.byref(fmt), pcm.data_ptr(),
                None if num_samples is None else num_samples.data_ptr(), num_packets,
                None if seg_first is None else seg_first.data_ptr(), nseg, int(max_segment_packets),
                None if state is None else state.data_ptr(), 1 if state_in else 0,
                ws.data_ptr(), ws.numel(), bufs["out"].data_ptr(), bufs["out"].numel(),
                bufs["sizes"].data_ptr(), bufs["offsets"].data_ptr())
            self._check(rc)
            for x in bufs.values():
                if hasattr(x, "record_stream"):
                    x.record_stream(cur)
            return bufs

    def encode_to_host(self, fmt, pcm, num_packets, **kw):
        """Convenience for tests: returns (stream bytes ndarray, sizes ndarray)."""
        b = self.encode(fmt, pcm, num_packets, **kw)
        self.synchronize()
        total = int(b["offsets"][-1].item())
        return b["out"][:total].cpu().numpy(), b["sizes"].cpu().numpy().astype(np.uint32)

    def encode_host(self, fmt, pcm, total_samples, segment_packets=0, state=None):
        """alac_hip_encode_host: host PCM (uint8 ndarray, total_samples sample-frames) -> (stream ndarray, sizes ndarray,
        final state ndarray).  segment_packets = 0: ONE chained segment (a whole file); k: state reset every k packets.
        Synchronous (does its own H2D / D2H)."""
        pcm = np.ascontiguousarray(pcm, np.uint8)
        npk = (total_samples + fmt.frame_size - 1) // fmt.frame_size
        nseg = (npk + segment_packets - 1) // segment_packets if segment_packets else 1
        cap = int(self.lib.alac_hip_encode_max_output_bytes(C.byref(fmt), npk))
        out = np.zeros(cap, np.uint8)
        sizes = np.zeros(max(npk, 1), np.uint32)
        n16 = int(self.lib.alac_hip_state_int16(C.byref(fmt)))
        st = np.zeros(nseg * n16, np.int16) if state is None else np.ascontiguousarray(state, np.int16).copy()
        total = _u64(0)
        self._check(self.lib.alac_hip_encode_host(self.h, C.byref(fmt), pcm.ctypes.data, total_samples, segment_packets,
                                                  st.ctypes.data, 0 if state is None else 1, out.ctypes.data, cap,
                                                  sizes.ctypes.data, C.byref(total)))
        return out[:total.value].copy(), sizes[:npk].copy(), st

    def profile_begin(self, max_calls):
        self._check(self.lib.alac_hip_profile_begin(self.h, max_calls))

    def profile_end(self):
        """-> (calls, {stage name: (mean ms per launch, launches per call)}); synchronises."""
        n = _u32(0)
        ns = self.lib.alac_hip_num_stages()
        ms = (C.c_float * ns)()
        ln = (_u32 * ns)()
        self._check(self.lib.alac_hip_profile_end(self.h, C.byref(n), ms, ln))
        return n.value, {self.lib.alac_hip_stage_name(i).decode(): (float(ms[i]), int(ln[i])) for i in range(ns)}

    def magic_cookie(self, fmt, max_frame_bytes=0, avg_bit_rate=0):
        c = np.zeros(48, np.uint8)
        n = self.lib.alac_hip_magic_cookie_full(C.byref(fmt), max_frame_bytes, avg_bit_rate, c.ctypes.data, 48)
        assert n == self.lib.alac_hip_magic_cookie_size(C.byref(fmt)) and n in (24, 48)
        return c[:n].copy()

    # ---- decode ----------------------------------------------------------------------------
    def decode(self, cookie, stream, offsets, num_packets, zero_fill=True, out=None):
        """stream: uint8 cuda tensor, offsets: int64 cuda tensor [num_packets+1].
        Returns (pcm uint8 tensor [num_packets*packet_bytes], num_samples int32, status int32).
        zero_fill=False leaves the bytes behind a short packet's samples uninitialised (the library writes
        num_samples frames per packet and zeroes only failed packets) and saves a pass over the output.
        out=(pcm, num_samples, status): write into these tensors instead of allocating (a timed loop that allocates
        2 GB per call measures the allocator)."""
        with self._call() as cur:
            t = self.torch
            ck = np.ascontiguousarray(cookie, np.uint8)
            fmt = Format()
            self._check(self.lib.alac_hip_format_from_cookie(ck.ctypes.data, ck.size, C.byref(fmt)))
            if out is not None:
                pcm, ns, st = out
                if pcm.numel() < num_packets * fmt.packet_bytes or ns.numel() < num_packets or st.numel() < num_packets:
                    raise ValueError("decode: output tensors too small")
            else:
                alloc = t.zeros if zero_fill else t.empty
                pcm = alloc(num_packets * fmt.packet_bytes, dtype=t.uint8, device=self.device)
                ns = t.zeros(num_packets, dtype=t.int32, device=self.device)
                st = t.zeros(num_packets, dtype=t.int32, device=self.device)
            # sized from the stream actually handed over (ID_FIL / ID_DSE padding may exceed the regular bound)
            wsb = int(self.lib.alac_hip_decode_workspace_bytes_stream(C.byref(fmt), num_packets, int(stream.numel())))
            ws = self._workspace(wsb)
            rc = self.lib.alac_hip_decode(self.h, ck.ctypes.data, ck.size, stream.data_ptr(), offsets.data_ptr(),
                                          num_packets, ws.data_ptr(), ws.numel(), pcm.data_ptr(), ns.data_ptr(),
                                          st.data_ptr())
            self._check(rc)
            for x in (pcm, ns, st):
                x.record_stream(cur)  # allocated on self.stream, consumed on the caller's
            return pcm, ns, st, fmt

    # ---- stage level ------------------------------------------------------------------------
    def pc_block(self, x, num, coefs, numactive, chanbits, denshift=9, decode=False):
        """x: int32 cuda [rows, stride]; coefs: int16 cuda [rows, 32] (adapted in place)."""
        with self._call() as cur:
            t = self.torch
            out = t.zeros_like(x)
            fn = self.lib.alac_hip_unpc_block if decode else self.lib.alac_hip_pc_block
            self._check(fn(self.h, x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], num,
                           coefs.data_ptr(), numactive, chanbits, denshift))
            out.record_stream(cur)
            return out

    def dyn_comp(self, pc, num_samples, bit_size, bytes_stride, mb0=10, pb=40, kb=14):
        with self._call() as cur:
            t = self.torch
            rows = pc.shape[0]
            bits = t.zeros((rows, bytes_stride), dtype=t.uint8, device=self.device)
            nb = t.zeros(rows, dtype=t.int32, device=self.device)
            self._check(self.lib.alac_hip_dyn_comp(self.h, mb0, pb, kb, pc.data_ptr(), rows, pc.shape[1],
                                                   num_samples, bit_size, bits.data_ptr(), bytes_stride,
                                                   nb.data_ptr()))
            bits.record_stream(cur)
            nb.record_stream(cur)
            return bits, nb

    def dyn_decomp(self, bits, num_samples, max_size, mb0=10, pb=40, kb=14):
        with self._call() as cur:
            t = self.torch
            rows, stride = bits.shape
            pc = t.zeros((rows, max(num_samples, 1)), dtype=t.int32, device=self.device)
            nb = t.zeros(rows, dtype=t.int32, device=self.device)
            st = t.zeros(rows, dtype=t.int32, device=self.device)
            self._check(self.lib.alac_hip_dyn_decomp(self.h, mb0, pb, kb, bits.data_ptr(), stride, rows,
                                                     pc.data_ptr(), pc.shape[1], num_samples, max_size,
                                                     nb.data_ptr(), st.data_ptr()))
            for x in (pc, nb, st):
                x.record_stream(cur)  # allocated on self.stream, consumed on the caller's
            return pc, nb, st


class Comm:
    """One alac_hip_comm (RCCL communicator of the re-assembly, include/alac_hip.h): one per process / GPU.

    `unique_id()` on rank 0 -> bytes every rank passes to the constructor (distributed by the caller's launcher channel).
    begin() / finish() enqueue on the torch stream that is current when they are called and never touch torch.distributed."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        rc = load_library().alac_hip_comm_unique_id(buf)
        if rc != 0:
            raise AlacError(rc, "alac_hip_comm_unique_id (librccl missing?)")
        return bytes(buf)

    def __init__(self, device, unique_id, rank, world):
        import torch
        self.torch = torch
        self.lib = load_library()
        self.device = torch.device("cuda", device)
        h = _vp()
        idb = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        rc = self.lib.alac_hip_comm_create(C.byref(h), device, idb, rank, world)
        if rc != 0:
            raise AlacError(rc, "alac_hip_comm_create failed")
        self.h, self.rank, self.world = h, rank, world

    def close(self):
        if getattr(self, "h", None):
            self.lib.alac_hip_comm_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc != 0:
            raise AlacError(rc, self.lib.alac_hip_comm_last_error(self.h).decode())

    def _stream(self):
        return _vp(self.torch.cuda.current_stream(self.device).cuda_stream)

    def begin(self, slot, offsets, num_packets, shard_capacity, out_capacity, sizes=None, all_sizes=None):
        """offsets: the int64 [num_packets + 1] tensor alac_hip_encode wrote (its last entry is the shard's byte count)"""
        self._check(self.lib.alac_hip_reassemble_begin(
            self.h, slot, offsets.data_ptr() + 8 * num_packets, int(shard_capacity), int(out_capacity),
            None if sizes is None else sizes.data_ptr(), num_packets, None if all_sizes is None else all_sizes.data_ptr(),
            self._stream()))

    def finish(self, slot, shard, stream_out):
        """-> byte offsets of the shards in stream_out, [world + 1]"""
        offs = (_u64 * (self.world + 1))()
        self._check(self.lib.alac_hip_reassemble_finish(self.h, slot, shard.data_ptr(), stream_out.data_ptr(), offs, self._stream()))
        return list(offs)
