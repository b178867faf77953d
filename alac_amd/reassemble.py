"""Re-assembly of a sharded ALAC stream across ranks.

Packets are byte-aligned (codec/ALACEncoder.cu:1039 in the reference), so concatenating the shards of
consecutive packet ranges is pure byte placement: rank r's shard goes at the sum of the sizes of the
shards before it.  The exchange is
  1. an all-gather of the shard byte counts (8 bytes per rank) -> the prefix-sum offsets,
  2. optionally an all-gather of the per-packet sizes (equal counts per rank: the CAF 'pakt' table),
  3. one GROUPED send/receive: every rank posts a receive for each peer's shard straight at that shard's final
     offset in the stream buffer and a send of its own shard to each peer — an all-gather with per-rank counts,
     no padding to the longest shard, no staging buffer, no second copy (only the rank's own shard is a local copy).
Works on any torch.distributed backend ("nccl" = RCCL over xGMI on the GPU node: the group becomes one
ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd; "gloo" on CPU for the tests).  xGMI is point to point, so every rank
receives its 7 peers' shards over 7 different links at once.

Since round 4 the PRODUCT's form of this exchange is C++ on librccl behind the C-ABI (alac_hip_comm_* / alac_hip_reassemble_*,
alac_amd/csrc/alac_comm.cpp; bench.py uses that one).  This module is the same protocol on torch.distributed: it is what the
CPU suite can run (gloo, world 2 and 3) and it keeps the padded all-gather as a second exchange mode for A/B runs.
"""
import torch
import torch.distributed as dist


def _all_gather_flat(out, piece, group, gloo):
    """out: [world * n] tensor, piece: [n] tensor of the same dtype"""
    if gloo:
        world = dist.get_world_size(group)
        parts = [torch.empty_like(piece) for _ in range(world)]
        dist.all_gather(parts, piece.contiguous(), group=group)
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, piece.contiguous(), group=group)


class Reassembler:
    """Two-phase exchange for a pipelined caller (bench.py at N > 1): begin() enqueues the small collectives (shard
    lengths and capacities, optionally packet sizes) and an asynchronous copy of the lengths to pinned host memory — no
    host wait; finish(), called a step later, reads the lengths (long since arrived), posts the grouped send/receive of
    the shard bytes at their final offsets.  The host therefore never blocks on the GPU between two encode steps.
    All work is issued on the stream that is current when the methods are called.

    The exchange MODE is a collective decision taken once, at construction: "direct" (grouped send/receive) unless any
    rank asks for the padded all-gather (mode="allgather" or ALAC_REASSEMBLE=allgather on that rank) — the request is
    all-reduced (MAX), so every rank runs the same sequence of collectives even if the ranks' environments differ.  A
    failing exchange raises on the rank that sees it; nothing falls back to another collective from an except branch
    (ranks that disagree about the next collective hang the job or place bytes wrongly).  Every precondition that can
    raise in finish() is evaluated from the all-gathered (length, capacity) table, identical on all ranks, so the ranks
    raise together, before any collective of that step."""

    def __init__(self, group=None, mode=None):
        import os
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.gloo = dist.get_backend(group) == "gloo"
        self.stream = None
        want = mode if mode is not None else os.environ.get("ALAC_REASSEMBLE", "")
        if want not in ("", "direct", "allgather"):
            raise ValueError(f"unknown re-assembly mode {want!r}")
        flag = torch.tensor([1 if want == "allgather" else 0], dtype=torch.int32,
                            device="cpu" if self.gloo else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        self.padded_mode = bool(flag.item())
        self.padded = None
        self._caps = {}

    def begin(self, shard, length, sizes=None):
        """shard: 1-D uint8 tensor whose first `length` bytes are this rank's packets; length: int64 tensor with one
        element on the shard's device; sizes: optional int32 tensor [packets per rank] (the same count on every rank)."""
        dev = shard.device
        # the capacity as a device tensor cached per (device, size): torch.tensor(n, device=cuda) is a synchronous pageable
        # host-to-device copy, and begin() must not wait for the GPU
        key = (str(dev), shard.numel())
        cap = self._caps.get(key)
        if cap is None:
            cap = self._caps[key] = torch.full((), shard.numel(), dtype=torch.int64, device=dev)
        mine = torch.stack([length.reshape(()).to(torch.int64), cap])
        table = torch.empty(2 * self.world, dtype=torch.int64, device=dev)
        _all_gather_flat(table, mine, self.group, self.gloo)
        table = table.view(self.world, 2)
        lens = table[:, 0].contiguous()
        all_sizes = None
        if sizes is not None:
            all_sizes = torch.empty(self.world * sizes.numel(), dtype=sizes.dtype, device=dev)
            _all_gather_flat(all_sizes, sizes, self.group, self.gloo)
        if dev.type == "cuda":
            table_h = torch.empty((self.world, 2), dtype=torch.int64, pin_memory=True)
            table_h.copy_(table, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        else:
            table_h, ready = table.clone(), None
        return dict(shard=shard, lens=lens, table_h=table_h, ready=ready, sizes=all_sizes)

    def finish(self, h):
        if h["ready"] is not None:
            h["ready"].synchronize()  # the lengths were exchanged a step ago
        table_h, shard = h["table_h"], h["shard"]
        lens_h, caps_h = table_h[:, 0], table_h[:, 1]
        dev = shard.device
        # preconditions, from the gathered table only: every rank reaches the same verdict
        if bool((lens_h > caps_h).any()) or bool((lens_h < 0).any()):
            raise ValueError("a shard buffer is shorter than its declared length")
        pad = max((int(lens_h.max().item()) + 15) // 16 * 16, 16)
        if self.padded_mode and pad > int(caps_h.min().item()):
            raise ValueError("padded all-gather: the longest shard rounded up to 16 bytes does not fit every rank's shard "
                             "buffer")
        mine = int(lens_h[self.rank].item())
        from .capi import shard_offsets  # the C-ABI's prefix sums (alac_hip_shard_offsets)
        offsets_h = torch.tensor(shard_offsets(lens_h.tolist()), dtype=torch.int64)
        total = int(offsets_h[-1].item())
        if self.stream is None or self.stream.numel() < total:
            self.stream = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        if self.padded_mode:
            self._place_padded(shard, lens_h, offsets_h, pad)
        else:
            self._place_direct(shard, lens_h, offsets_h, mine)
        return dict(stream=self.stream, total=total, lens=h["lens"], offsets=offsets_h, sizes=h["sizes"],
                    mode="padded all-gather" if self.padded_mode else "grouped send/recv")

    def _place_direct(self, shard, lens_h, offsets_h, mine):
        ops = []
        for r in range(self.world):
            n, o = int(lens_h[r].item()), int(offsets_h[r].item())
            if r == self.rank:
                self.stream[o:o + n].copy_(shard[:n])
            elif n > 0:
                ops.append(dist.P2POp(dist.irecv, self.stream[o:o + n], self._peer(r), group=self.group))
        if mine > 0:
            for r in range(self.world):
                if r != self.rank:
                    ops.append(dist.P2POp(dist.isend, shard[:mine], self._peer(r), group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()  # NCCL: orders the current stream behind the group; gloo: blocks until it has completed

    def _place_padded(self, shard, lens_h, offsets_h, pad):
        if self.padded is None or self.padded.numel() != self.world * pad:
            self.padded = torch.empty(self.world * pad, dtype=torch.uint8, device=shard.device)
        _all_gather_flat(self.padded, shard[:pad], self.group, self.gloo)
        for r in range(self.world):
            n, o = int(lens_h[r].item()), int(offsets_h[r].item())
            self.stream[o:o + n].copy_(self.padded[r * pad:r * pad + n])

    def _peer(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)


def reassemble_shards(shard, length, group=None, cache=None, sizes=None, mode=None):
    """One-shot form: returns dict(stream=<uint8 tensor, all shards in rank order>, total, lens=<int64[world]>,
    offsets=<int64[world+1]>, sizes).  `cache` (a previous return value) lets the stream buffer be reused."""
    if cache and mode is not None and (mode == "allgather") != cache["_ra"].padded_mode:
        raise ValueError("reassemble_shards: `mode` differs from the mode the cached Reassembler was constructed with "
                         "(the mode is a collective decision taken once, at construction)")
    ra = cache["_ra"] if cache else Reassembler(group, mode)
    out = ra.finish(ra.begin(shard, length, sizes))
    out["_ra"] = ra
    return out
