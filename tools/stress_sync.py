"""Repeat the fused (producer/consumer) encode and decode many times and compare every result on the device with
the first one (which bench.py / the tests check against the oracle): any lost hand-off would show as a mismatch."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import alac_amd

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    fmt = alac_amd.make_format(4096, 16, 2)
    ctx = alac_amd.Context(0)
    d_pcm = torch.from_numpy(alac_amd.synth_pcm(0, n, fmt)).cuda()
    ref = ctx.encode(fmt, d_pcm, n)
    ctx.synchronize()
    total = int(ref["offsets"][-1].item())
    ref_out = ref["out"][:total].clone(); ref_sizes = ref["sizes"].clone()
    cookie = ctx.magic_cookie(fmt)
    bad_e = bad_d = 0
    t0 = time.time()
    for i in range(iters):
        b = ctx.encode(fmt, d_pcm, n)
        out, ns, st, _ = ctx.decode(cookie, ref_out, ref["offsets"], n)
        ctx.synchronize()
        if not (torch.equal(b["out"][:total], ref_out) and torch.equal(b["sizes"], ref_sizes)):
            bad_e += 1
        if not (torch.equal(out, d_pcm) and int(st.abs().sum()) == 0):
            bad_d += 1
        if (i + 1) % 50 == 0:
            print(f"iter {i+1}: encode mismatches {bad_e}, decode mismatches {bad_d}, {time.time()-t0:.1f} s", flush=True)
    print("RESULT", "ok" if bad_e == 0 and bad_d == 0 else "MISMATCH", bad_e, bad_d)
    sys.exit(0 if bad_e == 0 and bad_d == 0 else 1)
main()
