"""Time the stage-level pc_block for deep predictors (BASELINE configs[2] 'deep LPC'): tap-parallel kernel vs the
lane-serial one (ALAC_HIP_STAGE_TAPS=0 in a second process).  usage: stage_taps_timing.py [rows=20000] [num=4096]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import alac_amd


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    num = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    ctx = alac_amd.Context(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    t = torch.arange(num + 64, device="cuda").float()
    x = (20000 * torch.sin(t[None, :] * (0.01 + 0.2 * torch.rand(rows, 1, device="cuda", generator=g)))
         + torch.randint(-2000, 2000, (rows, num + 64), device="cuda", generator=g)).to(torch.int32).contiguous()
    for na in (8, 16, 24, 30):
        co = torch.zeros((rows, 32), dtype=torch.int16, device="cuda")
        co[:, :3] = torch.tensor([1216, -928, -64], dtype=torch.int16)
        best = 1e9
        for _ in range(3):
            c = co.clone()
            torch.cuda.synchronize(); t0 = time.time()
            ctx.pc_block(x, num, c, na, 17)
            ctx.synchronize(); torch.cuda.synchronize()
            best = min(best, time.time() - t0)
        print(f"taps={na:2d} rows={rows} num={num}: {best*1e3:.2f} ms = {rows*num/best/1e9:.2f} G residuals/s "
              f"({'lane-serial' if os.environ.get('ALAC_HIP_STAGE_TAPS') == '0' else 'tap-parallel'} above 4 taps)")


main()
