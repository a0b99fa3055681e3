"""Multi-GPU data parallelism for independent utterances.

Utterances share no state (the reference's generator is per request, lm/generate.py:25-57), so
the path shards by utterance: one process per GPU, every rank holds a full replica and its own
slots; there is no per-step exchange.  The only collectives are a start-up broadcast of the packed
weight arena from rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests) and small reductions of
counters / timings for reporting.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def dist_env() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment (1 process => 0,1,0)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(backend: Optional[str] = None, device_index: Optional[int] = None) -> Tuple[int, int, int]:
    """``device_index``: the GPU this rank uses when it is not LOCAL_RANK (a rank started under a one-device
    HIP_VISIBLE_DEVICES mask uses device 0); returned in place of LOCAL_RANK."""
    rank, world, local = dist_env()
    if device_index is not None:
        local = device_index
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SMOLTTS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_utterances(n: int, rank: int, world: int) -> List[int]:
    """Round-robin utterance -> rank map ``u mod world``."""
    return list(range(rank, n, world))


def broadcast_weights(arena: Optional[torch.Tensor], offsets, device, src: int = 0):
    """Rank ``src`` passes its packed arena (any device) and offsets; every rank returns
    (arena on ``device``, offsets).  One message for the metadata, one for the bytes."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return arena.to(device), offsets
    rank = dist.get_rank()
    meta = [offsets, int(arena.numel())] if rank == src else [None, None]
    dist.broadcast_object_list(meta, src=src)
    offsets, nbytes = meta
    if dist.get_backend() == "gloo":  # CPU transport (tests, single-GPU rehearsals): stage through host memory
        buf = arena.cpu() if rank == src else torch.empty(nbytes, dtype=torch.uint8)
        dist.broadcast(buf, src=src)
        return buf.to(device), offsets
    buf = arena.to(device) if rank == src else torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src)
    return buf, offsets


def arena_checksum(arena: torch.Tensor) -> Tuple[float, float, float]:
    """Three order-sensitive-enough sums of a packed byte arena, each exact in float64 (they travel through the float
    all-gather): the byte sum and two 20-bit fields of its int32 words.  Computed where the arena lives, in pieces."""
    a = arena.reshape(-1)
    n4 = (a.numel() // 4) * 4
    s0 = s1 = s2 = 0
    step = 1 << 26  # bytes per piece: bounds the int64 temporaries
    for i in range(0, a.numel(), step):
        s0 += int(a[i:i + step].to(torch.int64).sum())
    for i in range(0, n4, step):
        w = a[i:min(i + step, n4)].view(torch.int32).to(torch.int64)
        s1 += int((w & 0xFFFFF).sum())
        s2 += int(((w >> 12) & 0xFFFFF).sum())
    return float(s0), float(s1), float(s2)


def identical_on_all_ranks(values: Sequence[float], device) -> bool:
    """True when every rank holds the same tuple of numbers (e.g. ``arena_checksum``): all-gathered and compared on every rank."""
    ok = True
    for v in values:
        got = all_gather_floats(float(v), device)
        ok = ok and len(set(got)) == 1
    return ok


def _reduce_device(device):
    return "cpu" if dist.get_backend() == "gloo" else device


def all_reduce_max(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def ranks_seen(device) -> int:
    """How many ranks the collective backend really connects: an all-reduce (sum) of one 1 per rank -- over RCCL on the
    rank's GPU, over gloo on the CPU.  1 without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.float32, device=_reduce_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def all_gather_floats(value: float, device) -> List[float]:
    """Every rank's ``value``, in rank order (reporting only)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=_reduce_device(device))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":  # RCCL: name the rank's own GPU instead of letting the collective guess it
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
