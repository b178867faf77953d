"""Worker of tests/test_gpu_comm.py::test_three_ranks_over_the_mock: WORLD threads of this process are the ranks of one job
on ONE GPU; libalac_hip.so resolves its RCCL calls in tests/cpp/libmock_rccl.so (ALAC_HIP_RCCL_LIB, set by the test).
Every rank encodes its own shard, then all run alac_hip_reassemble_begin / _finish; every rank must end up with the whole
stream = the shards in rank order, the gathered packet-size table and the shard offsets."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import alac_amd  # noqa: E402
from alac_amd.capi import AlacError, Comm  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    packets = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [300] * world
    fmt = alac_amd.make_format(512, 16, 2)
    uid = Comm.unique_id()
    B = max(packets)
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            ctx = alac_amd.Context(0)
            n = packets[r]  # ragged shards: every rank its own packet count (the size table needs equal counts: only then gathered)
            first = sum(packets[:r])
            d_pcm = ctx.synth_pcm(first, n, fmt)
            b = ctx.encode(fmt, d_pcm, n)
            ctx.synchronize()
            total = int(b["offsets"][-1].item())
            comm = Comm(0, uid, r, world)
            cap = int(ctx.lib.alac_hip_encode_max_output_bytes(fmt, B))
            out = torch.full((world * cap,), 0x5A, dtype=torch.uint8, device="cuda")
            equal = len(set(packets)) == 1
            all_sizes = torch.zeros(world * n, dtype=torch.int32, device="cuda")
            for slot in (0, 1):  # twice: slots and buffers are reusable
                if equal:
                    comm.begin(slot, b["offsets"], n, b["out"].numel(), out.numel(), sizes=b["sizes"], all_sizes=all_sizes)
                else:
                    comm.begin(slot, b["offsets"], n, b["out"].numel(), out.numel())
                offs = comm.finish(slot, b["out"], out)
                torch.cuda.synchronize()
            results[r] = dict(shard=b["out"][:total].cpu().numpy(), sizes=b["sizes"].cpu().numpy(), offs=offs,
                              stream=out[:offs[-1]].cpu().numpy(), rest_untouched=bool((out[offs[-1]:] == 0x5A).all()),
                              all_sizes=all_sizes.cpu().numpy())
            if len(sys.argv) > 3 and sys.argv[3] == "refuse":
                # one rank declares an output buffer that is too small: EVERY rank must refuse, nobody may post the group
                cap = out.numel() if r != 1 else offs[-1] - 1
                comm.begin(2, b["offsets"], n, b["out"].numel(), cap)
                try:
                    comm.finish(2, b["out"], out)
                    results[r]["refused"] = False
                except AlacError as e:
                    results[r]["refused"] = "does not fit" in str(e)
            comm.close()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    if errors or any(t.is_alive() for t in threads) or any(x is None for x in results):
        print("FAILED", errors, [t.is_alive() for t in threads])
        return 1
    want = np.concatenate([x["shard"] for x in results])
    lens = [len(x["shard"]) for x in results]
    offs = [0] + list(np.cumsum(lens))
    for r, x in enumerate(results):
        assert x["offs"] == [int(o) for o in offs], (r, x["offs"], offs)
        assert np.array_equal(x["stream"], want), f"rank {r}: re-assembled stream differs"
        assert x["rest_untouched"], f"rank {r}: bytes behind the stream were written"
        if len(set(packets)) == 1:
            assert np.array_equal(x["all_sizes"], np.concatenate([y["sizes"] for y in results])), f"rank {r}: size table"
        if "refused" in x:
            assert x["refused"], f"rank {r} did not refuse"
    print("OK", world, lens)
    return 0


if __name__ == "__main__":
    sys.exit(main())
