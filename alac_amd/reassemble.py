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
    lengths, optionally packet sizes) and an asynchronous copy of the lengths to pinned host memory — no host wait;
    finish(), called a step later, reads the lengths (long since arrived), posts the grouped send/receive of the shard
    bytes at their final offsets.  The host therefore never blocks on the GPU between two encode steps.
    All work is issued on the stream that is current when the methods are called."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.gloo = dist.get_backend(group) == "gloo"
        self.stream = None
        # ALAC_REASSEMBLE=allgather (or a grouped send/receive that raises) falls back to the padded all-gather: every
        # shard padded to the longest, one collective, then one copy per shard to its offset
        import os
        self.padded_mode = os.environ.get("ALAC_REASSEMBLE", "") == "allgather"
        self.padded = None

    def begin(self, shard, length, sizes=None):
        """shard: 1-D uint8 tensor whose first `length` bytes are this rank's packets; length: int64 tensor with one
        element on the shard's device; sizes: optional int32 tensor [packets per rank] (the same count on every rank)."""
        dev = shard.device
        lens = torch.empty(self.world, dtype=torch.int64, device=dev)
        _all_gather_flat(lens, length.reshape(1).to(torch.int64), self.group, self.gloo)
        all_sizes = None
        if sizes is not None:
            all_sizes = torch.empty(self.world * sizes.numel(), dtype=sizes.dtype, device=dev)
            _all_gather_flat(all_sizes, sizes, self.group, self.gloo)
        if dev.type == "cuda":
            lens_h = torch.empty(self.world, dtype=torch.int64, pin_memory=True)
            lens_h.copy_(lens, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        else:
            lens_h, ready = lens.clone(), None
        return dict(shard=shard, lens=lens, lens_h=lens_h, ready=ready, sizes=all_sizes)

    def finish(self, h):
        if h["ready"] is not None:
            h["ready"].synchronize()  # the lengths were exchanged a step ago
        lens_h, shard = h["lens_h"], h["shard"]
        dev = shard.device
        mine = int(lens_h[self.rank].item())
        if mine > shard.numel():
            raise ValueError("shard buffer shorter than its declared length")
        from .capi import shard_offsets  # the C-ABI's prefix sums (alac_hip_shard_offsets)
        offsets_h = torch.tensor(shard_offsets(lens_h.tolist()), dtype=torch.int64)
        total = int(offsets_h[-1].item())
        if self.stream is None or self.stream.numel() < total:
            self.stream = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        if not self.padded_mode:
            try:
                self._place_direct(shard, lens_h, offsets_h, mine)
            except RuntimeError as e:  # a backend without grouped point-to-point: keep the job alive on the collective
                import sys
                print(f"alac_amd.reassemble: grouped send/receive failed ({e}); falling back to the padded all-gather",
                      file=sys.stderr)
                self.padded_mode = True
        if self.padded_mode:
            self._place_padded(shard, lens_h, offsets_h)
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

    def _place_padded(self, shard, lens_h, offsets_h):
        pad = max((int(lens_h.max().item()) + 15) // 16 * 16, 16)
        if pad > shard.numel():
            raise ValueError("shard buffer shorter than the longest shard (padded all-gather)")
        if self.padded is None or self.padded.numel() != self.world * pad:
            self.padded = torch.empty(self.world * pad, dtype=torch.uint8, device=shard.device)
        _all_gather_flat(self.padded, shard[:pad], self.group, self.gloo)
        for r in range(self.world):
            n, o = int(lens_h[r].item()), int(offsets_h[r].item())
            self.stream[o:o + n].copy_(self.padded[r * pad:r * pad + n])

    def _peer(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)


def reassemble_shards(shard, length, group=None, cache=None, sizes=None):
    """One-shot form: returns dict(stream=<uint8 tensor, all shards in rank order>, total, lens=<int64[world]>,
    offsets=<int64[world+1]>, sizes).  `cache` (a previous return value) lets the stream buffer be reused."""
    ra = cache["_ra"] if cache else Reassembler(group)
    out = ra.finish(ra.begin(shard, length, sizes))
    out["_ra"] = ra
    return out
