"""Developer aid: encode synthetic packets on the GPU and report per-packet differences vs the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import alac_amd
from oracle_lib import Oracle

def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    fmt = alac_amd.make_format(4096, depth, ch)
    pcm = alac_amd.synth_pcm(0, n, fmt)
    ctx = alac_amd.Context(0)
    t = time.time()
    stream, sizes = ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n)
    print("gpu encode wall %.3f s" % (time.time() - t))
    o = Oracle()
    enc = o.encoder(4096, depth, ch)
    off = 0
    bad = 0
    for p in range(n):
        enc.reset()
        pk = enc.encode_packet(pcm[p * fmt.packet_bytes:(p + 1) * fmt.packet_bytes], 4096)
        info = enc.last_info()
        g = stream[off:off + sizes[p]]
        ok = len(pk) == sizes[p] and np.array_equal(pk, g)
        if not ok:
            bad += 1
            m = min(len(pk), len(g))
            diff = np.nonzero(pk[:m] != g[:m])[0]
            print(p, "MISMATCH sizes", len(pk), sizes[p], "first diff byte", diff[:1], info,
                  "ref", pk[:24].tobytes().hex(), "gpu", g[:24].tobytes().hex())
        off += int(sizes[p])
    print("packets", n, "bad", bad, "total bytes", off)

main()
