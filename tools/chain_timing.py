"""Time one chained segment (a single file's worth of packets) and many of them side by side."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import alac_amd

def main():
    fmt = alac_amd.make_format(4096, 16, 2)
    ctx = alac_amd.Context(0)
    for nseg, per in ((1, 237), (1, 32), (64, 32), (1024, 32), (10000, 1)):
        n = nseg * per
        pcm = torch.from_numpy(alac_amd.synth_pcm(0, n, fmt)).cuda()
        seg = torch.arange(0, n + 1, per, dtype=torch.int32).cuda()
        bufs = ctx.encode_buffers(fmt, n)
        for it in range(2):
            torch.cuda.synchronize(); t = time.time()
            ctx.encode(fmt, pcm, n, seg_first=seg, bufs=bufs)
            torch.cuda.synchronize(); dt = time.time() - t
        print(f"segments {nseg:6d} x {per:4d} packets: {dt*1e3:9.2f} ms  = {dt/per*1e3:7.3f} ms per packet position, "
              f"{n*4096/dt/1e6:9.1f} Msamples/s")
def real_audio():
    """the reference's audio/50.wav (fixture tests/golden/wav50_pcm.xz) as ONE chained segment"""
    import lzma
    path = os.path.join(ROOT, "tests", "golden", "wav50_pcm.xz")
    if not os.path.exists(path):
        return
    pcm = np.frombuffer(lzma.decompress(open(path, "rb").read()), np.uint8)
    fmt = alac_amd.make_format(4096, 16, 2)
    n = pcm.size // fmt.packet_bytes  # whole packets only
    ctx = alac_amd.Context(0)
    d = torch.from_numpy(pcm[:n * fmt.packet_bytes].copy()).cuda()
    seg = torch.tensor([0, n], dtype=torch.int32).cuda()
    bufs = ctx.encode_buffers(fmt, n)
    for it in range(2):
        torch.cuda.synchronize(); t = time.time()
        ctx.encode(fmt, d, n, seg_first=seg, bufs=bufs)
        torch.cuda.synchronize(); dt = time.time() - t
    print(f"50.wav, {n} full packets chained: {dt*1e3:9.2f} ms  = {dt/n*1e3:7.3f} ms per packet position, "
          f"{n*4096/dt/1e6:9.1f} Msamples/s = {n*4096/dt/44100:6.1f} x real time")


main()
real_audio()
