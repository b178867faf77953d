"""Time the decode direction (BASELINE configs[4]): 10k packets encoded on the GPU, decoded back, checked."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import alac_amd

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    fmt = alac_amd.make_format(4096, depth, 2)
    ctx = alac_amd.Context(0)
    pcm = alac_amd.synth_pcm(0, n, fmt)
    d_pcm = torch.from_numpy(pcm).cuda()
    b = ctx.encode(fmt, d_pcm, n)
    ctx.synchronize()
    cookie = ctx.magic_cookie(fmt)
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        out, ns, st, _ = ctx.decode(cookie, b["out"], b["offsets"], n)
        ctx.synchronize(); torch.cuda.synchronize(); dt = time.time() - t
        print(f"decode {n} packets: {dt*1e3:.2f} ms = {n*4096/dt/1e6:.1f} Msamples/s")
    print("round trip ok:", bool(torch.equal(out, d_pcm)), "status zero:", int(st.abs().sum()) == 0)
main()
