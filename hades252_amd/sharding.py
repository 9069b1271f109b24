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


def strong_shard(rank: int, world: int, n_total: int) -> Tuple[int, int]:
    """Strong scaling: `n_total` permutations in all (BASELINE configs[4]: 2^30), rank g owns the contiguous range
    [g n / W, (g + 1) n / W) -- the split of SURVEY.md section 8(e); sizes differ by at most one."""
    return shard_range(rank, world, n_total)


def init_process_group(backend: str):
    """Rendezvous on the launcher's MASTER_ADDR / MASTER_PORT.  A job of several ranks must be told its port (every rank
    has to name the same one: torchrun and bench.py's own launcher both export it); a lone rank picks a free port
    itself, so that two single-rank jobs on one host never collide on a hard-coded default."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if int(os.environ.get("WORLD_SIZE", "1")) > 1:
                raise RuntimeError("MASTER_PORT is not set: a multi-rank job gets it from its launcher "
                                   "(torch.distributed.run --master-port P, or `python bench.py --gpus N`)")
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
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


def shutdown(device=None) -> None:
    """Last collective of a job: a barrier, then the process group is destroyed on every rank (so that ranks which are
    done may exit while rank 0 goes on alone, without a communicator left half-open)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        barrier(device)
        dist.destroy_process_group()


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


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def backend_name() -> str:
    """"nccl" (= RCCL on ROCm), "gloo", or "none" when no process group exists (world size 1)."""
    import torch.distributed as dist
    return str(dist.get_backend()) if dist.is_available() and dist.is_initialized() else "none"


def all_gather_tensor(t):
    """Every rank's copy of `t` (same shape and dtype on all ranks), in rank order, on t's device.  RCCL gathers device
    tensors in place; gloo (the CPU tests, and the one-GPU rehearsal of the multi-rank path) gets host copies.  This is
    the Merkle path's only exchange step: world x sub-trees-per-rank x 32 bytes."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    if dist.get_backend() == "gloo" and t.device.type != "cpu":
        host = t.detach().cpu().contiguous()
        outs = [torch.empty_like(host) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, host)
        return [o.to(t.device) for o in outs]
    src = t.contiguous()
    outs = [torch.empty_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, src)
    return outs


def gather_strings(text: str, device="cpu", width: int = 96) -> List[str]:
    """Every rank's short ASCII string, in rank order (bookkeeping: which physical device a rank sat on)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [text]
    raw = text.encode("ascii", "replace")[:width].ljust(width, b"\0")
    t = torch.tensor(list(raw), dtype=torch.int64, device=_comm_device(device))
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [bytes(int(v) for v in o.cpu().tolist()).rstrip(b"\0").decode("ascii", "replace") for o in outs]


def device_identity(torch, index: int) -> str:
    """What tells two physical GPUs apart: PCI address (domain:bus:device.function) and, where the runtime exposes it, the
    device UUID.  From torch's device properties only: asking the HIP runtime directly would mean dlopen-ing a
    libamdhip64 by name, which need not be the copy torch has mapped (a second runtime in the rank process).  Without a
    PCI address the answer is "unknown device N", which `distinct_devices` treats as NOT distinct."""
    parts = []
    try:
        p = torch.cuda.get_device_properties(index)
        dom, bus, dev = (getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is not None:
            parts.append("pci %04x:%02x:%02x.0" % (int(dom or 0), int(bus), int(dev or 0)))
        uuid = getattr(p, "uuid", None)
        if uuid is not None:
            parts.append("uuid %s" % uuid)
        parts.append(str(getattr(p, "name", "")))
    except Exception:
        pass
    return "; ".join(x for x in parts if x) or "unknown device %d" % index


def distinct_devices(idents: List[str]) -> bool:
    """True when no two ranks report the same physical device (compared by PCI address / UUID, not by name)."""
    keys = []
    for s in idents:
        key = [x for x in s.split("; ") if x.startswith(("pci ", "uuid "))]
        keys.append(tuple(key) if key else (s,))
    return len(set(keys)) == len(keys)

