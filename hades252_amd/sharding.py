"""Multi-GPU plumbing: the batch partitions into independent permutations, so the only
cross-rank traffic is bookkeeping (a barrier, a max over elapsed times, a sum of digests).
There is no data-path collective.

One process per GPU; ``torch.distributed`` with backend "nccl" (RCCL) on GPUs, "gloo" in the
CPU tests.  The permutation of a state depends on nothing but that state
(reference src/strategies.rs:140-157 touches only ``data``), so a contiguous range per rank is
a complete decomposition.
"""
from __future__ import annotations

import os
from typing import List, Tuple

M64 = (1 << 64) - 1


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(rank: int, world: int, n_total: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of permutation indices owned by `rank` (balanced to +-1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return n_total * rank // world, n_total * (rank + 1) // world


def weak_shard(rank: int, per_rank: int) -> Tuple[int, int]:
    """Weak scaling: every rank owns `per_rank` permutations; global index range of `rank`."""
    return rank * per_rank, (rank + 1) * per_rank


def init_process_group(backend: str):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend)
    return dist


def _comm_device(device):
    """Tensors handed to collectives live on the GPU for nccl (RCCL) and on the host for gloo."""
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        return "cpu"
    return device


def barrier(device=None) -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() != "gloo" and device is not None and getattr(device, "type", "cpu") == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def reduce_max(value: float, device="cpu") -> float:
    """MAX over ranks of a python float (elapsed seconds)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum_int(value: int, device="cpu") -> int:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.int64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def reduce_min_int(value: int, device="cpu") -> int:
    """MIN over ranks (used as a logical AND of per-rank verification results)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.int64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def gather_floats(value: float, device="cpu") -> List[float]:
    """Every rank's value, in rank order (bookkeeping only: per-GPU kernel times)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def combine_digests(digest4: List[int], device="cpu") -> List[int]:
    """Digests are additive over disjoint index ranges (include/hades252.h): the digest of the
    whole job is the limb-wise wrapping sum of the shard digests.  Sent as 8 x 32-bit halves so
    the int64 all-reduce cannot overflow for world sizes < 2^31."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [d & M64 for d in digest4]
    halves = []
    for d in digest4:
        halves += [d & 0xFFFFFFFF, (d >> 32) & 0xFFFFFFFF]
    t = torch.tensor(halves, dtype=torch.int64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    h = [int(x) for x in t.cpu().tolist()]
    return [(h[2 * k] + (h[2 * k + 1] << 32)) & M64 for k in range(4)]
