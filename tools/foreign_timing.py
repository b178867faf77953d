"""Decode rate of a stream from ANOTHER encoder (orders 4..6, a denominator shift per packet — ffmpeg's defaults) against a
stream of this library's encoder of the same shape:  python tools/foreign_timing.py [PACKETS] [FRAME]
(500 distinct forged packets are repeated to fill the batch: the decoder does not care)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import alac_amd  # noqa: E402
import forge  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    frame = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    o, rng = Oracle(), np.random.default_rng(1)
    f = forge.Forger(o)
    distinct, src = [], []
    for i in range(500):
        pcm = forge.test_signal(rng, int(rng.integers(1, 3)), frame, 16, 2, headroom_bits=1)
        params = []
        for _ in range(2):
            num, den = int(rng.choice([4, 5, 6])), int(rng.choice([6, 7, 8, 9]))
            cp = forge.ChannelParams(num, den, 4, 0)
            cp.coefs[:num] = forge.default_coefs(num, den)[:num]
            params.append(cp)
        distinct.append(f.element(pcm, frame, 16, 2, frame, params, mix_bits=2, mix_res=int(rng.integers(0, 5))))
        src.append(np.frombuffer(bytes(pcm), np.uint8))
    pk = [distinct[i % 500] for i in range(n)]
    want = np.concatenate([src[i % 500] for i in range(n)])
    ctx = alac_amd.Context(0)
    fmt = alac_amd.make_format(frame, 16, 2)
    stream = torch.from_numpy(np.concatenate(pk)).cuda()
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum([len(x) for x in pk])]).astype(np.int64)).cuda()
    ck = forge.cookie(frame, 16, 2)
    outb = (torch.empty(n * fmt.packet_bytes, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
            torch.zeros(n, dtype=torch.int32, device="cuda"))

    def timed(cookie, s, of, label):
        with torch.cuda.stream(ctx.stream):
            ctx.decode(cookie, s, of, n, out=outb)
            ctx.synchronize()
            best = 1e9
            for _ in range(4):
                t = time.perf_counter()
                for _ in range(5):
                    ctx.decode(cookie, s, of, n, out=outb)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t) / 5)
        print(f"{label}: {best * 1e3:.3f} ms per {n} packets of {frame} = {n * frame / best / 1e9:.1f} Gsamples/s", flush=True)

    timed(ck, stream, offs, "foreign stream (orders 4..6, denShift 6..9)")
    assert int(outb[2].abs().sum()) == 0 and np.array_equal(outb[0].cpu().numpy(), want), "foreign decode differs from the source"
    d_pcm = torch.from_numpy(want).cuda()
    b = ctx.encode(fmt, d_pcm, n)
    ctx.synchronize()
    timed(ctx.magic_cookie(fmt), b["out"], b["offsets"], "the same PCM from this library's encoder")
    assert torch.equal(outb[0], d_pcm)


if __name__ == "__main__":
    main()
