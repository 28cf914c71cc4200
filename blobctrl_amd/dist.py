"""Multi-GPU: one process per GPU, independent edit requests sharded across ranks, frozen weights broadcast once.

The reference has no distributed code at all: every edit request is independent, so the path shards with
NO data-path collective.  The only exchange is at start-up: rank 0 packs the weights (LoRA merged, fp16 layouts) and
broadcasts the two arenas of each trunk over RCCL/xGMI (`torch.distributed.broadcast`, backend "nccl" = RCCL on ROCm),
so a node loads / converts a checkpoint once instead of eight times.  Works unchanged on the gloo backend (CPU tests).
"""
import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from .weights import PackedTrunk


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_requests(num_requests: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of independent edit requests to ranks (no cross-request state)."""
    return list(range(rank, num_requests, world))


def broadcast_packed(build_on_src: Callable[[], PackedTrunk], device, src: int = 0) -> PackedTrunk:
    """Rank `src` builds the packed trunk; everyone receives an identical replica (2 broadcasts of the arenas)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return build_on_src()
    rank = dist.get_rank()
    pw = build_on_src() if rank == src else None
    meta = [pw.meta() if rank == src else None]
    dist.broadcast_object_list(meta, src=src)
    if rank != src:
        pw = PackedTrunk.from_meta(meta[0], device)
    dist.broadcast(pw.h_arena, src=src)
    dist.broadcast(pw.f_arena, src=src)
    return pw


def gather_results(local: Sequence[torch.Tensor], dst: int = 0):
    """Optional: collect final latents (32 KiB per 512x512 edit) on one rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(local)]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object([t.cpu() for t in local], out, dst=dst)
    return out


def barrier_max_seconds(seconds: float, device=None) -> float:
    """MAX over ranks of a per-rank wall time (bench contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def arena_hash(*trunks: PackedTrunk) -> str:
    """16 hex digits over the weight arenas of the given trunks as this rank holds them (two wrapping 64-bit sums - of the words and of
    the words times their position modulo a prime - computed on the device in chunks): equal on every rank after `broadcast_packed`."""
    s1 = s2 = 0
    mask = (1 << 64) - 1
    for pw in trunks:
        for arena in (pw.h_arena, pw.f_arena):
            raw = arena.view(torch.int16) if arena.dtype == torch.float16 else arena.view(torch.int32)
            step = 1 << 24
            for o in range(0, raw.numel(), step):
                x = raw[o:o + step].to(torch.int64)
                pos = (torch.arange(x.numel(), device=x.device, dtype=torch.int64) + o) % 65521 + 1
                s1 = (s1 + int(x.sum().item())) & mask
                s2 = (s2 + int((x * pos).sum().item())) & mask
    return f"{s1 & 0xFFFFFFFF:08x}{s2 & 0xFFFFFFFF:08x}"
