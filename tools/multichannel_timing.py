"""Time encode and decode of a stream of 3..8 channels next to the stereo path on the same number of
channel-chains.  usage: multichannel_timing.py [channels=6] [packets=3334] [depth=16]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import alac_amd


def interleave(parts, bps):
    cols = [np.ascontiguousarray(p, np.uint8).reshape(-1, c * bps) for p, c in parts]
    return np.concatenate(cols, axis=1).reshape(-1)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize()
        best = min(best, time.time() - t)
    return best


def main():
    ch = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3334
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    bps = {16: 2, 20: 3, 24: 3, 32: 4}[depth]
    ctx = alac_amd.Context(0)
    fmt = alac_amd.make_format(4096, depth, ch)
    # channel pairs from the stereo generator, a last odd channel from the mono one
    parts, c, k = [], 0, 0
    while c < ch:
        w = 2 if c + 2 <= ch else 1
        parts.append((alac_amd.synth_pcm(k * n, n, alac_amd.make_format(4096, depth, w)), w))
        c += w; k += 1
    pcm = interleave(parts, bps)
    d_pcm = torch.from_numpy(pcm).cuda()
    bufs = ctx.encode_buffers(fmt, n)
    te = timed(lambda: (ctx.encode(fmt, d_pcm, n, bufs=bufs), ctx.synchronize()))
    cookie = ctx.magic_cookie(fmt)
    b = ctx.encode(fmt, d_pcm, n, bufs=bufs); ctx.synchronize()
    res = {}
    def dec():
        res["o"] = ctx.decode(cookie, b["out"], b["offsets"], n); ctx.synchronize()
    td = timed(dec, 3)
    out, ns, st, _ = res["o"]
    ok = bool(torch.equal(out, d_pcm)) and int(st.abs().sum()) == 0
    total = int(b["offsets"][-1].item())
    print(f"{ch} ch {depth}-bit, {n} packets ({n*ch} channel-chains), {total/len(pcm):.3f} of the PCM size")
    print(f"  encode {te*1e3:.2f} ms = {n*4096/te/1e6:.1f} Msamples/s ({n*4096*ch/te/1e9:.2f} G channel-samples/s)")
    print(f"  decode {td*1e3:.2f} ms = {n*4096/td/1e6:.1f} Msamples/s, round trip ok: {ok}")
    # the stereo path on the same number of chains
    f2 = alac_amd.make_format(4096, depth, 2)
    n2 = n * ch // 2
    p2 = torch.from_numpy(alac_amd.synth_pcm(0, n2, f2)).cuda()
    b2 = ctx.encode_buffers(f2, n2)
    t2 = timed(lambda: (ctx.encode(f2, p2, n2, bufs=b2), ctx.synchronize()))
    ck2 = ctx.magic_cookie(f2)
    bb = ctx.encode(f2, p2, n2, bufs=b2); ctx.synchronize()
    t2d = timed(lambda: (ctx.decode(ck2, bb["out"], bb["offsets"], n2), ctx.synchronize()), 3)
    print(f"  stereo, {n2} packets: encode {t2*1e3:.2f} ms, decode {t2d*1e3:.2f} ms")


main()
