#!/usr/bin/env python3
"""Where and when the single-wave workgroups of the fused final launch ran (option "debug_waves"):

    python tools/wave_map.py PACKETS "k=v,k=v" ["k=v" ...]

For every option set: one encode pass with stamps, then per wave kind (predictor / coder) the distribution of durations and
the number of SIMDs that hosted 0 / 1 / 2 / 3+ waves of the launch (HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13;
XCC_ID 3:0)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

import alac_amd  # noqa: E402


def main():
    B = int(sys.argv[1])
    fmt = alac_amd.make_format(4096, 16, 2, 44100)
    ctx = alac_amd.Context(0)
    d_pcm = ctx.synth_pcm(0, B, fmt)
    bufs = ctx.encode_buffers(fmt, B)
    off = int(ctx.lib.alac_hip_debug_waves_offset(C.byref(fmt), B, B))
    nLms, nC = (2 * B + 31) // 32, (2 * B + 63) // 64
    nW = 3 * nC  # workers of the launch: roles dealt P P C (k_final_fused)
    is_coder = np.arange(nW) % 3 == 2
    exists = np.where(is_coder, True, (np.arange(nW) // 3) * 2 + np.arange(nW) % 3 < nLms)
    for s in sys.argv[2:]:
        opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",") if kv)
        with ctx.options(debug_waves=1, **opts):
            for _ in range(6):
                ctx.encode(fmt, d_pcm, B, bufs=bufs)
            ctx.synchronize()
        w = ctx._ws[off:off + 32 * nW].cpu().numpy().view(np.uint32).reshape(-1, 8)[exists]
        coder = is_coder[exists]
        hw, xcc = w[:, 0], w[:, 1] & 15
        t0 = w[:, 2].astype(np.uint64) | (w[:, 3].astype(np.uint64) << 32)
        t1 = w[:, 4].astype(np.uint64) | (w[:, 5].astype(np.uint64) << 32)
        dur = (t1 - t0).astype(np.float64)
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        key = [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i])) for i in range(len(w))]
        cnt = collections.Counter(key)
        per_cu = collections.Counter(k[:4] for k in key)
        span = float(t1.max() - t0.min())
        print(f"{opts}: launch span {span:.0f} shader-clock ticks")
        for name, sl in (("predictor", ~coder), ("coder", coder)):
            d = dur[sl]
            print(f"   {name:9s} waves {len(d):4d}  duration ticks min {d.min():.0f} median {np.median(d):.0f} p90 {np.percentile(d, 90):.0f} max {d.max():.0f}"
                  f"   start spread {float(t0[sl].max() - t0[sl].min()):.0f}")
        print("   waves per SIMD histogram:", dict(sorted(collections.Counter(cnt.values()).items())), " SIMDs used", len(cnt),
              " waves per CU histogram:", dict(sorted(collections.Counter(per_cu.values()).items())), " CUs used", len(per_cu))
        # do shared SIMDs explain slow waves?
        shared = np.array([cnt[k] > 1 for k in key])
        coders_per_cu = collections.Counter(k[:4] for k, c in zip(key, coder) if c)
        print("   coder waves per CU histogram:", dict(sorted(collections.Counter(coders_per_cu.values()).items())))
        for name, sl in (("predictor", ~coder), ("coder", coder)):
            d, sh_ = dur[sl], shared[sl]
            if sh_.any() and (~sh_).any():
                print(f"   {name}: alone on its SIMD median {np.median(d[~sh_]):.0f}, sharing {np.median(d[sh_]):.0f} ({int(sh_.sum())} waves share)")
        moved = int((w[:, 0] != w[:, 6]).sum())
        print(f"   waves whose HW_ID changed between entry and exit: {moved}")


if __name__ == "__main__":
    main()
