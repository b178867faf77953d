"""Chained segments side by side (the shape of `alacconvert --batch`: files x packets) at segment counts around the edges of the
four-lane / two-lane windows (v1_narrow_regime): automatic choice against both forced forms.  GPU box: python tools/chain_sweep.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import alac_amd
fmt = alac_amd.make_format(4096, 16, 2)
ctx = alac_amd.Context(0)
for nseg, per in ((3000, 8), (5000, 8), (6000, 8), (8000, 6), (12000, 4), (16000, 4)):
    n = nseg * per
    pcm = ctx.synth_pcm(0, n, fmt)
    seg = torch.arange(0, n + 1, per, dtype=torch.int32).cuda()
    bufs = ctx.encode_buffers(fmt, n)
    res = {}
    for name, opts in (("auto", {}), ("narrow=0", {"narrow": 0}), ("narrow=1", {"narrow": 1})):
        with ctx.options(**opts):
            best = 1e9
            for it in range(4):
                torch.cuda.synchronize(); t = time.time()
                ctx.encode(fmt, pcm, n, seg_first=seg, bufs=bufs)
                torch.cuda.synchronize(); best = min(best, time.time() - t)
        res[name] = best * 1e3
    print(f"segments {nseg:6d} x {per} packets: " + "  ".join(f"{k} {v:8.3f} ms" for k, v in res.items()), flush=True)
