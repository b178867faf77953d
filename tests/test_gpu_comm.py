"""GPU: the RCCL re-assembly behind the C-ABI (include/alac_hip.h alac_hip_comm_* / alac_hip_reassemble_*; alac_comm.cpp)
with ONE rank — every RCCL call of the path runs (ncclGetUniqueId, ncclCommInitRank, both ncclAllGather, the table read-back;
the send/receive group has no peers at world 1), the shard lands at offset 0 and the gathered size table is the rank's own.
World > 1 runs the SAME C++ (alac_comm.cpp) with its RCCL calls resolved in a test double (tests/cpp/mock_rccl.cpp: ranks are
threads of one process on the one GPU; RCCL itself refuses two ranks on a device): placement at the prefix-sum offsets,
ragged shards, the group's send/receive matching, the all-ranks-alike refusal.  tests/test_reassemble_gloo.py keeps the
torch.distributed form of the same exchange under gloo."""
import os
import subprocess
import sys
import numpy as np
import pytest

import alac_amd
from alac_amd.capi import AlacError, Comm

pytestmark = pytest.mark.gpu


def test_single_rank_reassembly_through_the_c_abi(gpu_ctx):
    import torch
    fmt = alac_amd.make_format(1024, 16, 2)
    n = 600
    d_pcm = gpu_ctx.synth_pcm(0, n, fmt)
    b = gpu_ctx.encode(fmt, d_pcm, n)
    gpu_ctx.synchronize()
    total = int(b["offsets"][-1].item())
    comm = Comm(0, Comm.unique_id(), 0, 1)
    out = torch.full((total + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    all_sizes = torch.zeros(n, dtype=torch.int32, device="cuda")
    for slot in (0, 3, 1):  # any slot, repeatedly
        out.fill_(0x5A)
        comm.begin(slot, b["offsets"], n, b["out"].numel(), out.numel(), sizes=b["sizes"], all_sizes=all_sizes)
        offs = comm.finish(slot, b["out"], out)
        torch.cuda.synchronize()
        assert offs == [0, total]
        assert torch.equal(out[:total], b["out"][:total]) and bool((out[total:] == 0x5A).all())
        assert torch.equal(all_sizes, b["sizes"])
    # a pipelined caller: two passes outstanding on different slots, finished in order
    comm.begin(0, b["offsets"], n, b["out"].numel(), out.numel())
    comm.begin(1, b["offsets"], n, b["out"].numel(), out.numel())
    assert comm.finish(0, b["out"], out) == [0, total] and comm.finish(1, b["out"], out) == [0, total]
    # refusals: a busy slot, a finish without begin, an output buffer that is too small, a shard longer than its buffer
    comm.begin(2, b["offsets"], n, b["out"].numel(), out.numel())
    with pytest.raises(AlacError):
        comm.begin(2, b["offsets"], n, b["out"].numel(), out.numel())
    comm.finish(2, b["out"], out)
    with pytest.raises(AlacError):
        comm.finish(2, b["out"], out)
    comm.begin(0, b["offsets"], n, b["out"].numel(), total - 1)
    with pytest.raises(AlacError) as ei:
        comm.finish(0, b["out"], out)
    assert ei.value.code == -50 and "does not fit" in str(ei.value)
    comm.begin(0, b["offsets"], n, total - 1, out.numel())
    with pytest.raises(AlacError) as ei:
        comm.finish(0, b["out"], out)
    assert "shorter than its declared length" in str(ei.value)
    torch.cuda.synchronize()
    comm.close()


def test_comm_argument_checks():
    with pytest.raises(AlacError):
        Comm(0, bytes(128), 1, 1)  # rank >= world
    with pytest.raises(AlacError):
        Comm(0, bytes(128), 0, 0)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,packets,mode", [(3, "300,300,300", ""), (2, "64,200", ""), (4, "100,7,256,31", "refuse")])
def test_several_ranks_over_the_mock(gpu_ctx, world, packets, mode):
    """the product's N > 1 exchange, every RCCL call resolved in tests/cpp/libmock_rccl.so (its own process: the library binds
    librccl once)"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "libmock_rccl.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ALAC_HIP_RCCL_LIB=os.path.join(ROOT, "tests", "cpp", "libmock_rccl.so"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "comm_mock_worker.py"), str(world), packets, mode],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and p.stdout.strip().startswith("OK"), (p.stdout[-1500:], p.stderr[-1500:])
