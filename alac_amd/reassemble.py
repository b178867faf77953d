"""Re-assembly of a sharded ALAC stream across ranks.

Packets are byte-aligned (codec/ALACEncoder.cu:1039 in the reference), so concatenating the shards of
consecutive packet ranges is pure byte placement: rank r's shard goes at the sum of the sizes of the
shards before it.  Two collectives: an all-gather of the shard byte counts, then an all-gather of the
shard bytes padded to the longest shard.  Works on any torch.distributed backend ("nccl" = RCCL over
xGMI on the GPU node; "gloo" on CPU for the tests).
"""
import torch
import torch.distributed as dist


def reassemble_shards(shard, length, group=None, cache=None):
    """shard: 1-D uint8 tensor holding this rank's packed packets in its first `length` bytes.
    length: int64 tensor with one element (same device as shard).
    Returns dict(stream=<contiguous uint8 tensor, all shards in rank order>, lens=<int64[world]>,
    offsets=<int64[world+1]>).  `cache` (a previous return value) lets buffers be reused."""
    world = dist.get_world_size(group)
    dev = shard.device
    lens = cache["lens"] if cache else torch.empty(world, dtype=torch.int64, device=dev)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(parts, length.reshape(1).to(torch.int64), group=group)
        lens.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(lens, length.reshape(1).to(torch.int64), group=group)
    lens_h = lens.cpu()  # 8 x world bytes; the only host read of the exchange
    maxlen = int(lens_h.max().item())
    pad = (maxlen + 15) // 16 * 16
    if pad > shard.numel():
        raise ValueError("shard buffer shorter than its declared length")
    padded = cache.get("padded") if cache else None
    pad = max(pad, 16)
    if padded is None or padded.shape[1] != pad:
        padded = torch.empty((world, pad), dtype=torch.uint8, device=dev)
    if dist.get_backend(group) != "gloo":
        dist.all_gather_into_tensor(padded, shard[:pad], group=group)
    else:
        parts = [torch.empty(pad, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(parts, shard[:pad].contiguous(), group=group)
        for r in range(world):
            padded[r, :pad].copy_(parts[r])
    offsets_h = torch.zeros(world + 1, dtype=torch.int64)
    offsets_h[1:] = torch.cumsum(lens_h, 0)
    total = int(offsets_h[-1].item())
    stream = cache.get("stream") if cache else None
    if stream is None or stream.numel() < total:
        stream = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    for r in range(world):
        n = int(lens_h[r].item())
        o = int(offsets_h[r].item())
        stream[o:o + n].copy_(padded[r, :n])
    return dict(stream=stream, total=total, lens=lens, offsets=offsets_h, padded=padded)


class Reassembler:
    """Two-phase form of reassemble_shards for a pipelined caller (bench.py at N > 1): begin() enqueues the small
    exchange of shard lengths and an asynchronous copy of them to pinned host memory — no host wait; finish(),
    called a step later, reads the lengths (long since arrived), then enqueues the padded all-gather of the shard
    bytes and the byte placement.  The host therefore never blocks on the GPU between two encode steps.
    All work is issued on the stream that is current when the methods are called."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.gloo = dist.get_backend(group) == "gloo"
        self.padded = None
        self.stream = None

    def begin(self, shard, length):
        dev = shard.device
        lens = torch.empty(self.world, dtype=torch.int64, device=dev)
        if self.gloo:
            parts = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
            dist.all_gather(parts, length.reshape(1).to(torch.int64), group=self.group)
            lens.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(lens, length.reshape(1).to(torch.int64), group=self.group)
        if dev.type == "cuda":
            lens_h = torch.empty(self.world, dtype=torch.int64, pin_memory=True)
            lens_h.copy_(lens, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        else:
            lens_h, ready = lens.clone(), None
        return dict(shard=shard, lens=lens, lens_h=lens_h, ready=ready)

    def finish(self, h):
        if h["ready"] is not None:
            h["ready"].synchronize()  # the lengths were exchanged a step ago
        lens_h, shard = h["lens_h"], h["shard"]
        dev = shard.device
        pad = max((int(lens_h.max().item()) + 15) // 16 * 16, 16)
        if pad > shard.numel():
            raise ValueError("shard buffer shorter than its declared length")
        if self.padded is None or self.padded.shape[1] != pad:
            self.padded = torch.empty((self.world, pad), dtype=torch.uint8, device=dev)
        if not self.gloo:
            dist.all_gather_into_tensor(self.padded, shard[:pad], group=self.group)
        else:
            parts = [torch.empty(pad, dtype=torch.uint8, device=dev) for _ in range(self.world)]
            dist.all_gather(parts, shard[:pad].contiguous(), group=self.group)
            for r in range(self.world):
                self.padded[r, :pad].copy_(parts[r])
        offsets_h = torch.zeros(self.world + 1, dtype=torch.int64)
        offsets_h[1:] = torch.cumsum(lens_h, 0)
        total = int(offsets_h[-1].item())
        if self.stream is None or self.stream.numel() < total:
            self.stream = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        for r in range(self.world):
            n, o = int(lens_h[r].item()), int(offsets_h[r].item())
            self.stream[o:o + n].copy_(self.padded[r, :n])
        return dict(stream=self.stream, total=total, lens=h["lens"], offsets=offsets_h, padded=self.padded)
