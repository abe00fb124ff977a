"""Multi-GPU: one process per GPU, batch-of-prompts sharded by rank, ONE collective in the whole job.

The reference has no inference parallelism.  Images are independent chains, so the
denoise loop needs no exchange; the only communication is the start-up broadcast of the packed weight
arena (rank 0 loads + packs once; 2.46 GB bf16 for the 1.23 B UNet) over RCCL/xGMI.
torch.distributed is used as plumbing (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) slice of a batch of prompts for this rank."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_arena(arena: Optional[torch.Tensor], nbytes: int, device, src: int = 0, chunk_bytes: int = 1 << 30) -> torch.Tensor:
    """Rank `src` passes its packed arena; every other rank passes None and receives a copy.
    Sent as a few large chunks (xGMI rings are per-link bound; large messages amortise latency)."""
    if not (dist.is_available() and dist.is_initialized()):
        assert arena is not None
        return arena
    if dist.get_rank() != src:
        arena = torch.empty(nbytes, dtype=torch.uint8, device=device)
    assert arena.numel() == nbytes, "arena size mismatch between ranks"
    for o in range(0, nbytes, chunk_bytes):
        dist.broadcast(arena[o:min(nbytes, o + chunk_bytes)], src=src)
    return arena


def gather_outputs(local: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Optional final gather of per-rank results (e.g. uint8 images) to rank `dst`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    world = dist.get_world_size()
    sizes = [torch.zeros(1, dtype=torch.long, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.long, device=local.device))
    mx = int(max(s.item() for s in sizes))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if dist.get_rank() != dst:
        return None
    return [b[: int(s.item())] for b, s in zip(bufs, sizes)]
