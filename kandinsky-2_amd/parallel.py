"""Multi-GPU: one process per GPU, batch-of-prompts sharded by rank, ONE collective in the whole job.

The reference has no inference parallelism.  Images are independent chains, so the
denoise loop needs no exchange; the only communication is the start-up broadcast of the packed weight
arena (rank 0 loads + packs once; 2.46 GB bf16 for the 1.23 B UNet) over RCCL/xGMI.
torch.distributed is used as plumbing (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import os
import socket
from typing import Callable, List, Optional, Sequence

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


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_entry(rank: int, world: int, port: int, fn: Callable, args: Sequence):
    # exactly what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` exports
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    fn(rank, world, *args)


def launch_ranks(fn: Callable, world: int, args: Sequence = (), devices_visible: Optional[int] = None, timeout_s: float = 3600.0) -> None:
    """Self-launcher for `--gpus N` without torch.distributed.run: starts `world` processes (one per GPU) that run
    fn(rank, world, *args) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT exported.  Fails LOUDLY - never quietly
    runs fewer ranks - when fewer devices are visible than ranks, or when a rank exits non-zero."""
    if world < 1:
        raise ValueError("world must be >= 1")
    if devices_visible is not None and devices_visible < world:
        raise RuntimeError(f"{world} ranks requested but only {devices_visible} device(s) visible: refusing to run a smaller job under the same name")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_entry, args=(r, world, port, fn, tuple(args))) for r in range(world)]
    for p in procs:
        p.start()
    # ONE deadline for the whole job, all ranks polled together: the first rank that dies takes the others down at once (ranks that
    # lose a peer block in their next RCCL collective and would otherwise be waited for one timeout each)
    import time
    deadline = time.monotonic() + timeout_s
    bad = []
    while True:
        alive = [p for p in procs if p.is_alive()]
        bad = [(r, p.exitcode) for r, p in enumerate(procs) if not p.is_alive() and p.exitcode != 0]
        if bad or not alive:
            break
        if time.monotonic() > deadline:
            bad = [(r, "timeout") for r, p in enumerate(procs) if p.is_alive()]
            break
        time.sleep(0.05)
    if bad:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10.0)
            if p.is_alive():
                p.kill()
        raise RuntimeError(f"launch_ranks: ranks failed: {bad}")
    for p in procs:
        p.join()
