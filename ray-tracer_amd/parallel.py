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
