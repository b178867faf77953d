"""CPU, world_size 2 over gloo: the N > 1 path of bench.py (shard -> encode -> re-assemble) gives the
same stream as one process encoding everything.  Shards are encoded with the CPU oracle here (ragged shard lengths,
so the grouped send/receive places every shard at a different, unaligned offset); on the GPU node the same
Reassembler runs over RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, per_rank, q, mode=""):
    # "mixed": the ranks' environments disagree — the mode is still one collective decision (any rank's request wins)
    os.environ["ALAC_REASSEMBLE"] = ("allgather" if rank == 0 else "") if mode == "mixed" else mode
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import alac_amd
    from alac_amd.reassemble import Reassembler, reassemble_shards
    from oracle_lib import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fmt = alac_amd.make_format(4096, 16, 2)
    pcm = alac_amd.synth_pcm(rank * per_rank, per_rank, fmt)
    s, sizes = Oracle().encoder(4096, 16, 2).encode_stream(pcm, per_rank * 4096, 1)
    cap = per_rank * 16400
    shard = torch.zeros(cap, dtype=torch.uint8)
    shard[:len(s)] = torch.from_numpy(s)
    res = None
    t_sizes = torch.from_numpy(sizes.astype(np.int32))
    for _ in range(2):  # second call reuses the cached buffers
        res = reassemble_shards(shard, torch.tensor([len(s)], dtype=torch.int64), None, res, sizes=t_sizes)
    # the pipelined two-phase form (what bench.py uses at N > 1) must give the same stream
    ra = Reassembler()
    h1 = ra.begin(shard, torch.tensor([len(s)], dtype=torch.int64))
    h2 = ra.begin(shard, torch.tensor([len(s)], dtype=torch.int64))
    r1 = ra.finish(h1)
    first = r1["stream"][:r1["total"]].clone()
    r2 = ra.finish(h2)
    assert torch.equal(first, res["stream"][:res["total"]]) and torch.equal(r2["stream"][:r2["total"]], first)
    assert r1["mode"] == ("grouped send/recv" if mode == "" else "padded all-gather")
    q.put((rank, res["stream"][:res["total"]].numpy().copy(), res["offsets"].numpy().copy(), res["sizes"].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("WORLD,mode", [(2, ""), (3, ""), (2, "allgather"), (2, "mixed")])
def test_rank_reassembly_equals_single_stream(WORLD, mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import alac_amd
    from oracle_lib import Oracle
    world, per_rank = WORLD, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fmt = alac_amd.make_format(4096, 16, 2)
    pcm = alac_amd.synth_pcm(0, world * per_rank, fmt)
    want, sizes = Oracle().encoder(4096, 16, 2).encode_stream(pcm, world * per_rank * 4096, 1)
    for rank, stream, offsets, all_sizes in got:
        assert np.array_equal(stream, want), f"rank {rank}"
        assert offsets[-1] == len(want) and offsets[1] == int(sizes[:per_rank].sum())
        assert np.array_equal(all_sizes.astype(np.uint32), sizes)  # the 'pakt' table of the whole stream


def _worker_short_buffer(rank, world, port, q):
    """padded mode, rank 1's shard buffer too short for the longest shard: EVERY rank must raise, before any collective of
    the step (a ValueError on a subset of the ranks strands the others inside the all-gather)"""
    sys.path.insert(0, ROOT)
    from alac_amd.reassemble import Reassembler
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000 if rank == 0 else 100
    shard = torch.zeros(4096 if rank == 0 else 100, dtype=torch.uint8)  # rank 1: exactly its own length
    ra = Reassembler(mode="allgather" if rank == 1 else None)
    h = ra.begin(shard, torch.tensor([n], dtype=torch.int64))
    try:
        ra.finish(h)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, "raised: " + str(e)))
    dist.barrier()  # reached by both: nobody is stuck in a collective the other never entered
    # and a direct-mode exchange still works afterwards on the same group
    rb = Reassembler(mode="direct")
    out = rb.finish(rb.begin(shard, torch.tensor([n], dtype=torch.int64)))
    q.put((rank, int(out["total"])))
    dist.destroy_process_group()


def test_padded_capacity_failure_is_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_short_buffer, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    first = {r: v for r, v in got if isinstance(v, str)}
    assert set(first) == {0, 1} and all(v.startswith("raised: padded all-gather") for v in first.values()), got
    assert sorted(v for r, v in got if isinstance(v, int)) == [1100, 1100]
