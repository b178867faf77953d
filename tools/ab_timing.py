#!/usr/bin/env python3
"""A/B timing of code-path options on one shape, in ONE process (interleaved repeats):

    python tools/ab_timing.py PACKETS DEPTH "k=v,k=v" ["k=v,..." ...]   [--decode] [--rounds R] [--passes P]

Every option set encodes the same device-generated batch; prints ms per pass (host clock around P passes, best and median
of R rounds), the library's stage timing, and whether the bytes equal those of the FIRST option set and (sampled) the CPU
oracle.  "" = defaults."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import alac_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("packets", type=int)
    ap.add_argument("depth", type=int)
    ap.add_argument("sets", nargs="+")
    ap.add_argument("--decode", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--passes", type=int, default=10)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--state", action="store_true", help="pass a caller-owned coefficient state (k_init_state runs)")
    a = ap.parse_args()
    fmt = alac_amd.make_format(4096, a.depth, a.channels, 44100)
    B = a.packets
    ctx = alac_amd.Context(0)
    d_pcm = ctx.synth_pcm(0, B, fmt)
    sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",") if kv) for s in a.sets]
    bufs = [ctx.encode_buffers(fmt, B) for _ in sets]
    kw = {"state": torch.zeros((B, 64), dtype=torch.int16, device="cuda")} if a.state else {}
    times = [[] for _ in sets]
    stage = [None] * len(sets)
    cookie = ctx.magic_cookie(fmt)
    d_out = (torch.empty(B * fmt.packet_bytes, dtype=torch.uint8, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda"),
             torch.zeros(B, dtype=torch.int32, device="cuda"))
    with torch.cuda.stream(ctx.stream):
        for r in range(a.rounds + 1):
            for i, opts in enumerate(sets):
                with ctx.options(**opts):
                    if a.decode:
                        if r == 0:
                            ctx.encode(fmt, d_pcm, B, bufs=bufs[i], **kw)
                        ctx.decode(cookie, bufs[i]["out"], bufs[i]["offsets"], B, out=d_out)
                        ctx.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(a.passes):
                            ctx.decode(cookie, bufs[i]["out"], bufs[i]["offsets"], B, out=d_out)
                        ctx.synchronize()
                        dt = (time.perf_counter() - t0) / a.passes
                        if r == 0:
                            ok = bool(torch.equal(d_out[0], d_pcm)) and int(d_out[2].abs().sum()) == 0
                            print(f"set {i} {opts}: decode round trip exact: {ok}")
                    else:
                        ctx.encode(fmt, d_pcm, B, bufs=bufs[i], **kw)
                        ctx.synchronize()
                        if r == a.rounds:
                            ctx.profile_begin(a.passes)
                        t0 = time.perf_counter()
                        for _ in range(a.passes):
                            ctx.encode(fmt, d_pcm, B, bufs=bufs[i], **kw)
                        ctx.synchronize()
                        dt = (time.perf_counter() - t0) / a.passes
                        if r == a.rounds:
                            stage[i] = ctx.profile_end()[1]
                    if r > 0:
                        times[i].append(dt * 1e3)
    total0 = int(bufs[0]["offsets"][-1].item())
    for i, opts in enumerate(sets):
        same = True
        if not a.decode and i > 0:
            same = bool(torch.equal(bufs[i]["sizes"], bufs[0]["sizes"])) and bool(torch.equal(bufs[i]["out"][:total0], bufs[0]["out"][:total0]))
        t = times[i]
        line = f"set {i} {opts}: best {min(t):.4f} ms  median {statistics.median(t):.4f} ms  = {B * 4096 / min(t) / 1e6:.1f} Gsamples/s"
        if not a.decode:
            line += f"  same bytes as set 0: {same}"
            if stage[i]:
                line += "\n      " + "  ".join(f"{k} {v[0]:.3f}x{v[1]}" for k, v in stage[i].items() if v[0] > 0)
        print(line)
    if not a.decode and not a.no_oracle:
        from oracle_lib import Oracle
        enc = Oracle().encoder(4096, a.depth, 2, 44100)
        offs = bufs[0]["offsets"].cpu().numpy()
        idx = sorted(set(list(range(0, B, 997)) + [0, 1, B - 1]))
        ok = True
        for p in idx:
            enc.reset()
            want = enc.encode_packet(alac_amd.synth_pcm(p, 1, fmt), 4096)
            got = bufs[0]["out"][int(offs[p]):int(offs[p + 1])].cpu().numpy()
            ok = ok and np.array_equal(got, want)
        print(f"set 0 vs CPU oracle on {len(idx)} sampled packets: {ok}")


if __name__ == "__main__":
    main()
