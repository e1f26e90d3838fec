"""Row-interleaved film sharding across ranks (one process per GPU, torch.distributed).

Pixels are independent and the scene is read-only, so the only exchange is the film gather
at the end: rank r renders rows y with y % world == r into a packed [rows_r, W, 3] tensor and
rank 0 receives and re-interleaves them.  Interleaving (rather than contiguous stripes)
balances the very uneven per-pixel cost of glass / mirror regions.  The sampler is keyed on
the absolute pixel index, so the gathered film equals the single-rank film bit for bit.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(height: int, rank: int, world: int) -> int:
    return len(range(rank, height, world))


def gather_film(local: torch.Tensor, height: int, width: int, rank: int, world: int, dst: int = 0):
    """local: float32 [rows_of_rank, width, 3] on the rank's device.  Returns the full
    [height, width, 3] film on rank `dst`, None elsewhere.  world == 1: returns `local`."""
    assert local.dtype == torch.float32 and local.shape == (shard_rows(height, rank, world), width, 3)
    if world == 1:
        return local
    max_rows = shard_rows(height, 0, world)
    send = local
    if local.shape[0] != max_rows:  # ranks own either max_rows or max_rows - 1 rows: pad to a common size
        send = torch.zeros((max_rows, width, 3), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if rank == dst:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=parts, dst=dst)
        full = torch.empty((height, width, 3), dtype=local.dtype, device=local.device)
        for r in range(world):
            full[r::world] = parts[r][: shard_rows(height, r, world)]
        return full
    dist.gather(send, gather_list=None, dst=dst)
    return None


class ShardedFilms:
    """The shards of SEVERAL films (the scenes of one frame set) in ONE buffer per rank, so that a step makes one gather
    whatever the number of scenes, with nothing allocated inside the step.

    Every film's share is padded to the largest share (rows of rank 0), so all ranks send the same number of floats.
    `local[i]` is this rank's packed [rows, W, 3] view for film i: render into it.  `gather()` sends the buffer to rank 0, which
    receives into one [world, total] buffer and re-interleaves each film with one strided copy:
    packed row j of rank k is film row k + world * j."""

    def __init__(self, shapes, rank: int, world: int, device, via_host: bool = False, collective=None):
        # collective: whether gather() goes through torch.distributed at all; by default it does when there is more than one rank.
        # True with world == 1 runs the same calls in a group of one (a one-GPU machine can then exercise the RCCL path).
        self.collective = (world > 1) if collective is None else bool(collective)
        # via_host: the ranks meet over gloo (rehearsals on a machine with fewer GPUs than ranks): gloo takes host tensors only, so
        # the buffer is copied to host memory with a synchronous copy, gathered there, and copied back on rank 0.
        self.via_host = bool(via_host)
        self.shapes, self.rank, self.world = [(int(h), int(w)) for h, w in shapes], rank, world
        self.rmax = [shard_rows(h, 0, world) for h, _ in self.shapes]
        sizes = [r * w * 3 for r, (_, w) in zip(self.rmax, self.shapes)]
        self.offs = [sum(sizes[:i]) for i in range(len(sizes))]
        self.total = sum(sizes)
        self.send = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.local = []
        for (h, w), off in zip(self.shapes, self.offs):
            rows = shard_rows(h, rank, world)
            self.local.append(self.send[off: off + rows * w * 3].view(rows, w, 3))
        root = self.collective and rank == 0
        self.recv = torch.empty((world, self.total), dtype=torch.float32, device=device) if root else None
        self.full = [torch.empty((r * world, w, 3), dtype=torch.float32, device=device) for r, (_, w) in zip(self.rmax, self.shapes)] if root else None

    def bytes_into_root(self) -> int:
        return (self.world - 1) * self.total * 4

    def gather(self):
        """-> the whole films [H, W, 3] on rank 0 (views of buffers this object keeps), None on the other ranks."""
        if not self.collective:
            return list(self.local)
        if self.via_host:
            send = self.send.cpu()   # synchronous: the kernels that fill the buffer have finished when it returns
            if self.rank != 0:
                dist.gather(send, gather_list=None, dst=0)
                return None
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.gather(send, gather_list=parts, dst=0)
            self.recv.copy_(torch.stack(parts))
        elif self.rank != 0:
            dist.gather(self.send, gather_list=None, dst=0)
            return None
        else:
            dist.gather(self.send, gather_list=list(self.recv.unbind(0)), dst=0)
        films = []
        for (h, w), off, r, full in zip(self.shapes, self.offs, self.rmax, self.full):
            part = self.recv[:, off: off + r * w * 3].view(self.world, r, w, 3)
            full.view(r, self.world, w, 3).copy_(part.permute(1, 0, 2, 3))
            films.append(full[:h])
        return films
