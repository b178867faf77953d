"""forge.py — TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_golden.py, never by the product).

A packet FORGER: compressed ID_SCE / ID_LFE / ID_CPE elements with header parameters of the caller's choice — what
ANOTHER legal ALAC encoder could emit and what `ALACDecoder::Decode` therefore has to follow
(codec/ALACDecoder.cu:622-756 mono, :758-930 stereo): any `mixBits` / signed `mixRes` (:797-798), `mode` != 0 (the
first-order pass in front of the predictor, :829-838, :845-854), `denShift` 0..15 and `pbFactor` 0..7 (:800-812), `numU` /
`numV` 0..31 with arbitrary starting coefficients (:803-812), `bytesShifted` 0..2 (:772-777), partial frames (:783-787), any
instance tag, and cookies with other `pb` / `mb` / `kb` (`set_ag_params(&agParams, mConfig.mb, (pb * pbFactorU) / 4,
mConfig.kb, ...)`, :825, :841).  This library's own encoder only ever writes mode 0, denShift 9, pbFactor 4, 4 or 8 taps,
mixBits 2, mixRes 0..4 (codec/ALACEncoder.cu:466-485), so packets from `oracle.encoder()` cannot reach those decoder paths.

The forger is the ENCODER-side inverse of exactly the decoder's steps, built from the oracle's pinned stage functions
(`pc_block`, `dyn_comp`, `put_bits`; pass `stage_fns` to run it over the reference's compiled objects instead):

    PCM -> L, R (sign-extended; shifted-off low bytes to the shift buffer, codec/matrix_enc.cu:186-323)
        -> u = R + ((mixRes * v) >> mixBits), v = L - R            (the inverse of ALACDecoder.cu:193-223 for ANY mixRes;
                                                                      equal to mix16's (mixRes*L + (2^mixBits - mixRes)*R) >> mixBits)
        -> pc_block(u, coefs, num, chanBits, denShift)              (codec/dp_enc.c:77-388, general loop for other tap counts)
        -> [mode != 0: pc_block(.., 31)]                            (first difference: inverse of unpc_block(.., 31, ..))
        -> dyn_comp(mb, (pb * pbFactor) / 4, kb)                    (codec/ag_enc.c:249-367)

Losslessness holds whenever u, v fit `chanBits` signed bits (the caller picks amplitudes accordingly); where they do not,
the decoder's sign extension wraps identically in the oracle and on the GPU, which is what the parity tests compare.
"""
import ctypes as C

import numpy as np

BPS = {16: 2, 20: 3, 24: 3, 32: 4}


def cookie(frame_size, depth, channels, pb=40, mb=10, kb=14, max_run=255, rate=44100):
    """24-byte ALACSpecificConfig (codec/ALACAudioTypes.h:162-176), big-endian, with the caller's pb / mb / kb"""
    c = np.zeros(24, np.uint8)
    c[0:4] = np.frombuffer(int(frame_size).to_bytes(4, "big"), np.uint8)
    c[4] = 0
    c[5], c[6], c[7], c[8], c[9] = depth, pb, mb, kb, channels
    c[10:12] = np.frombuffer(int(max_run).to_bytes(2, "big"), np.uint8)
    c[20:24] = np.frombuffer(int(rate).to_bytes(4, "big"), np.uint8)
    return c


def pcm_to_channels(pcm, depth, channels, n):
    """packed little-endian interleaved PCM -> int64 [channels][n], right-aligned (20-bit: the top 20 of 24 bits)"""
    b = np.ascontiguousarray(pcm, np.uint8)[:n * channels * BPS[depth]].reshape(n, channels, BPS[depth]).astype(np.int64)
    if depth == 16:
        x = b[..., 0] | (b[..., 1] << 8)
        x = (x ^ 0x8000) - 0x8000
    elif depth == 32:
        x = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16) | (b[..., 3] << 24)
        x = (x ^ 0x80000000) - 0x80000000
    else:
        x = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
        x = (x ^ 0x800000) - 0x800000
        if depth == 20:
            x >>= 4
    return np.ascontiguousarray(x.T)


def channels_to_pcm(x, depth):
    """inverse of pcm_to_channels for values that fit the depth"""
    x = np.asarray(x, np.int64).T  # [n][channels]
    if depth == 20:
        x = x << 4
    nb = BPS[depth]
    out = np.zeros(x.shape + (nb,), np.uint8)
    for i in range(nb):
        out[..., i] = (x >> (8 * i)) & 0xff
    return out.reshape(-1)


class ChannelParams:
    """one channel's header bytes: (mode << 4 | denShift), (pbFactor << 5 | num), coefs[num]"""

    def __init__(self, num=8, den_shift=9, pb_factor=4, mode=0, coefs=None):
        self.num, self.den_shift, self.pb_factor, self.mode = num, den_shift, pb_factor, mode
        self.coefs = np.zeros(32, np.int16)
        if coefs is not None:
            self.coefs[:len(coefs)] = coefs


def default_coefs(num, den_shift):
    """init_coefs (codec/dp_enc.c:49-60) scaled to another denominator shift: a stable starting predictor"""
    c = np.zeros(32, np.int16)
    den = 1 << den_shift
    base = [(38 * den) >> 4, (-29 * den) >> 4, (-2 * den) >> 4]  # AINIT / BINIT / CINIT
    for i in range(min(num, 3)):
        c[i] = base[i]
    return c


class Forger:
    def __init__(self, oracle, stage_fns=None):
        """oracle: tests/oracle_lib.Oracle.  stage_fns: None (the oracle's own stage functions) or a dict with
        'pc_block', 'dyn_comp', 'put_bits' ctypes functions of the same flat signatures (oracle/ref_adapter.c)"""
        self.o = oracle
        f = stage_fns or {}
        self.pc_fn = f.get("pc_block")
        self.comp_fn = f.get("dyn_comp") or oracle.lib.oalac_dyn_comp
        self.put_fn = f.get("put_bits") or oracle.lib.oalac_put_bits

    def _put(self, buf, pos, value, nbits):
        self.put_fn(buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(pos), int(value) & 0xffffffff, nbits)

    def _residuals(self, x, n, cp, chan_bits):
        r, _ = self.o.pc_block(x, n, cp.coefs, cp.num, chan_bits, cp.den_shift, fn=self.pc_fn)
        if cp.mode != 0:
            r, _ = self.o.pc_block(r, n, np.zeros(32, np.int16), 31, chan_bits, 0, fn=self.pc_fn)
        return r[:n]

    def element(self, pcm, n, depth, channels, frame_size, params, mix_bits=0, mix_res=0, bytes_shifted=0,
                instance=0, pb=40, mb=10, kb=14, lfe=False, force_partial=False, end=True, buf=None, pos=None):
        """One compressed element for `channels` in (1, 2) from n sample-frames of packed PCM.  Returns the finished
        packet (element + ID_END + byte alignment) as bytes, or with end=False appends to (buf, pos) for multi-element
        packets.  params: [ChannelParams] per channel."""
        assert channels in (1, 2) and 0 <= bytes_shifted <= 2 and n <= frame_size
        x = pcm_to_channels(pcm, depth, channels, n)
        shift = 8 * bytes_shifted
        sh = x & ((1 << shift) - 1) if shift else None
        x = x >> shift
        chan_bits = depth - shift + (1 if channels == 2 else 0)
        if channels == 2:
            L, R = x[0], x[1]
            v = L - R
            u = R + ((mix_res * v) >> mix_bits) if mix_res != 0 else L
            if mix_res == 0:
                v = R
            planes = [u, v]
        else:
            planes = [x[0]]
        for p in planes:
            assert p.size == 0 or (p.min() >= -(1 << 31) and p.max() < (1 << 31))
        own = buf is None
        if own:
            buf = np.zeros(n * channels * 8 + 4096, np.uint8)
            pos = C.c_uint64(0)
        partial = 1 if (n != frame_size or force_partial) else 0
        self._put(buf, pos, (3 if lfe else 0) if channels == 1 else 1, 3)
        self._put(buf, pos, instance, 4)
        self._put(buf, pos, 0, 12)
        self._put(buf, pos, (partial << 3) | (bytes_shifted << 1), 4)
        if partial:
            self._put(buf, pos, n, 32)
        self._put(buf, pos, mix_bits, 8)
        self._put(buf, pos, mix_res & 0xff, 8)
        for cp in params:
            self._put(buf, pos, (cp.mode << 4) | cp.den_shift, 8)
            self._put(buf, pos, (cp.pb_factor << 5) | cp.num, 8)
            for i in range(cp.num):
                self._put(buf, pos, int(cp.coefs[i]) & 0xffff, 16)
        if shift:
            for z in range(n):  # interleaved shifted-off bytes, codec/ALACEncoder.cu:493-499
                for c in range(channels):
                    self._put(buf, pos, int(sh[c][z]), shift)
        for c, cp in enumerate(params):
            r = np.ascontiguousarray(self._residuals(planes[c].astype(np.int32), n, cp, chan_bits), np.int32)
            nbits = C.c_uint32(0)
            st = self.comp_fn(mb, (pb * cp.pb_factor) // 4, kb, r.ctypes.data_as(C.POINTER(C.c_int32)),
                              buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(pos), n, chan_bits, C.byref(nbits))
            assert st == 0, st
        if not end:
            return None
        return self.finish(buf, pos)

    def finish(self, buf, pos):
        """ID_END + byte alignment (codec/ALACEncoder.cu:1034-1039)"""
        self._put(buf, pos, 7, 3)
        return buf[:(pos.value + 7) // 8].copy()

    def raw_tag(self, tag, payload_bits=0):
        """a packet that starts with an arbitrary 3-bit element tag (ID_CCE = 2, ID_PCE = 5: kALAC_ParamError,
        codec/ALACDecoder.cu:932-939)"""
        buf = np.zeros(64, np.uint8)
        pos = C.c_uint64(0)
        self._put(buf, pos, tag, 3)
        self._put(buf, pos, 0x5a5a5a5a, 32)
        return buf[:8].copy()


def random_params(rng, num_choices=(0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 30, 31), den_choices=(4, 5, 6, 7, 8, 9, 10, 11, 12),
                  pb_choices=(1, 2, 3, 4, 5, 6, 7), mode_choices=(0, 0, 1)):
    num = int(rng.choice(num_choices))
    den = int(rng.choice(den_choices))
    cp = ChannelParams(num, den, int(rng.choice(pb_choices)), int(rng.choice(mode_choices)))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        cp.coefs[:] = default_coefs(num, den)
    elif kind == 1:
        cp.coefs[:num] = rng.integers(-(1 << den) // 2, (1 << den) // 2 + 1, size=num)
    else:
        cp.coefs[:num] = rng.integers(-32768, 32768, size=num)  # hostile: int16 range, exercises the wraps
    return cp


def test_signal(rng, kind, n, depth, channels, headroom_bits=2):
    """packed PCM whose samples leave `headroom_bits` of the depth unused (so that foreign mix weights stay lossless)"""
    lim = (1 << (depth - 1 - headroom_bits)) - 1
    t = np.arange(n)
    cols = []
    for c in range(channels):
        if kind == 0:
            x = rng.integers(-lim, lim + 1, size=n)
        elif kind == 1:
            x = np.cumsum(rng.integers(-lim // 64 - 2, lim // 64 + 3, size=n))
        elif kind == 2:
            x = 0.6 * lim * np.sin(t * (0.01 + 0.02 * c) + rng.random() * 6) + rng.integers(-8, 9, size=n)
        elif kind == 3:
            x = np.zeros(n)
            if n:
                k = max(n // 50, 1)
                x[rng.integers(0, n, size=k)] = rng.integers(-lim, lim + 1, size=k)
        else:
            x = rng.integers(-2, 3, size=n) * (rng.random(n) < 0.2)
        cols.append(np.clip(np.asarray(x).astype(np.int64), -lim - 1, lim))
    if channels == 2 and kind in (1, 2) and rng.random() < 0.5:
        cols[1] = np.clip(cols[0] + rng.integers(-20, 21, size=n), -lim - 1, lim)  # correlated pair: mixing matters
    return channels_to_pcm(np.stack(cols), depth)


def forge_batch(forger, rng, count, depth, channels, frame_size, pb=40, mb=10, kb=14, hostile=True):
    """`count` forged packets of one stream format (one cookie).  Returns (packets, pcm, lossless): packets[i] bytes,
    pcm[i] the packed source PCM of its n_i sample-frames, lossless[i] True where decode(packet) must equal pcm[i]
    (samples with headroom, a shift path the depth's output routine has, mixBits small enough for the headroom)."""
    packets, pcms, lossless = [], [], Flags()
    sizes = [frame_size, frame_size, frame_size // 2 + 3, 1, 2, 5, 17, 33, frame_size - 1, 100 % (frame_size + 1) or 1]
    for i in range(count):
        n = int(sizes[i % len(sizes)]) if i % 4 else frame_size
        n = max(1, min(n, frame_size))
        kind = int(rng.integers(0, 5))
        shifted = 0
        if depth >= 24 and rng.random() < 0.6:
            shifted = int(rng.integers(1, 3)) if depth == 32 else 1
        if depth == 32 and shifted == 0:
            shifted = 2 if channels == 2 else int(rng.integers(0, 3))  # stereo chanBits = 33 would not exist
        ok = True
        if depth in (16, 20) and hostile and rng.random() < 0.08:
            shifted = 1  # the decoder parses the shift bytes; its 16-/20-bit output routines ignore them (ALACDecoder.cu:193-280)
            ok = False
        mix_bits = int(rng.integers(0, 5))
        mix_res = int(rng.choice([0, 1, 2, 3, 4, -1, -2, -3, -7, 5, 9, 15, -16])) if channels == 2 else int(rng.integers(-128, 128))
        if channels == 1 and rng.random() < 0.5:
            mix_bits = int(rng.integers(0, 256))  # mono: read and ignored (:657-659)
        # full-scale samples wherever the mix is an interpolation (0 <= mixRes <= 2^mixBits: u lies between L and R), so
        # that predictor differences reach chanBits + 1 bits; extrapolating weights get the headroom they need
        headroom = 0 if rng.random() < 0.7 else 2
        if channels == 2 and (mix_res < 0 or mix_res > (1 << mix_bits)):
            headroom = 2 + int(np.ceil(np.log2(abs(mix_res) / (1 << mix_bits) + 1)))
        if depth - 8 * shifted - 1 - headroom < 2:
            headroom = max(depth - 8 * shifted - 3, 0)
            ok = ok and (channels == 1 or mix_res == 0 or abs(mix_res) <= (1 << mix_bits))
        pcm = test_signal(rng, kind, n, depth, channels, headroom_bits=headroom)
        params = [random_params(rng) if (hostile or rng.random() < 0.8) else ChannelParams(8, 9, 4, 0, default_coefs(8, 9))
                  for _ in range(channels)]
        if i % 7 == 3:  # a packet exactly like this library's encoder writes them, next to the foreign ones
            params = [ChannelParams(int(rng.choice([4, 8])), 9, 4, 0) for _ in range(channels)]
            for cp in params:
                cp.coefs[:] = default_coefs(cp.num, 9)
            if channels == 2:
                mix_bits, mix_res = 2, int(rng.integers(0, 5))
        pk = forger.element(pcm, n, depth, channels, frame_size, params, mix_bits=mix_bits if channels == 2 else mix_bits & 0xff,
                            mix_res=mix_res, bytes_shifted=shifted, instance=int(rng.integers(0, 16)), pb=pb, mb=mb, kb=kb,
                            lfe=(channels == 1 and rng.random() < 0.2), force_partial=bool(rng.random() < 0.1))
        packets.append(pk)
        pcms.append(pcm)
        lossless.append(ok)
        lossless.info.append(dict(n=n, shifted=shifted, mix=(mix_bits, mix_res), lfe=pk[0] >> 5 == 3,
                                  chans=[(cp.num, cp.den_shift, cp.pb_factor, cp.mode) for cp in params]))
    return packets, pcms, lossless


class Flags(list):
    """the lossless flags of forge_batch, with the packets' header parameters alongside (`info[i]`) for failure messages"""

    def __init__(self, *a):
        super().__init__(*a)
        self.info = []
