"""Multi-GPU plumbing of the scan path: one process per GPU, strings sharded by index, one tiny collective.

The path shards naturally (each string's walk is independent, pire/run.h:271-275), so there is NO data-path
collective: rank r owns the contiguous global string range shard_range(n_total, r, world) and scans it locally.
The only exchange is the sum of the uint64[regexps+2] match counters (80 B for 8 regexps) -- an all-reduce over
RCCL (torch.distributed backend "nccl" on ROCm) -- and, for timing, a MAX over ranks."""
from __future__ import annotations

import os
from typing import Tuple


# bench.py --force-dist: a world of ONE rank still sends its counters through the backend's all-reduce (with nccl: RCCL),
# so that a one-GPU box executes the collective path the 8-GPU run takes
FORCE_SINGLE_RANK_COLLECTIVES = False


def _collectives_on() -> bool:
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_SINGLE_RANK_COLLECTIVES)


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of global string indices owned by `rank` (sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def world_info() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device=None):
    """Initialise torch.distributed (rendezvous on 127.0.0.1 unless MASTER_ADDR says otherwise)."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")          # a world of one rank without a launcher (bench.py --force-dist)
    os.environ.setdefault("WORLD_SIZE", "1")
    if dist.is_initialized():
        return
    if device is not None and backend == "nccl":
        dist.init_process_group(backend=backend, device_id=device)
    else:
        dist.init_process_group(backend=backend)


def _gloo_with_device_tensor(t) -> bool:
    import torch.distributed as dist

    return dist.get_backend() == "gloo" and t.is_cuda


def allreduce_counts(counts, async_op: bool = False):
    """Sum the match counters over all ranks in place (no-op for a single process).

    async_op=True returns the collective's work handle (or None): the caller's stream is NOT made to wait for the
    reduction, so the next scan can run while RCCL moves its 80 bytes; call .wait() before touching `counts` again.
    With the gloo backend (control-flow tests on a box with fewer GPUs than ranks) device counters are reduced
    through the host, synchronously."""
    import torch.distributed as dist

    if _collectives_on():
        if _gloo_with_device_tensor(counts):
            host = counts.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            counts.copy_(host)
            return None if async_op else counts
        work = dist.all_reduce(counts, op=dist.ReduceOp.SUM, async_op=async_op)
        return work if async_op else counts
    return None if async_op else counts


def max_over_ranks(seconds: float, device=None) -> float:
    import torch
    import torch.distributed as dist

    if not _collectives_on():
        return seconds
    on = device if device is not None and dist.get_backend() != "gloo" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist

    if _collectives_on():
        dist.barrier()


def gather_over_ranks(value: float, device=None):
    """Every rank's `value`, in rank order, on every rank (one all-gather of a float64 each): the per-rank rates a
    scaling line is read against."""
    import torch
    import torch.distributed as dist

    if not _collectives_on():
        return [value]
    on = device if device is not None and dist.get_backend() != "gloo" else "cpu"
    mine = torch.tensor([value], dtype=torch.float64, device=on)
    out = [torch.zeros(1, dtype=torch.float64, device=on) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def backend_description() -> str:
    """What carries the counter all-reduce: "rccl <version> (torch.distributed backend nccl)", "gloo", or "none"."""
    import torch
    import torch.distributed as dist

    if not _collectives_on():
        return "none (one rank)"
    name = dist.get_backend()
    if name == "nccl":
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            ver = "?"
        return f"rccl {ver} (torch.distributed backend nccl, HIP {torch.version.hip})"
    return name


def rccl_version():
    """The RCCL this torch was built with ("2.26.6"), whether or not a process group is up; None without one."""
    try:
        import torch

        return ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:   # noqa: BLE001
        return None


def describe_ranks(device=None):
    """What every rank of the process group is and sits on, in rank order, on every rank: [{"rank", "host", "device", "pci_bus_id",
    "name"}] (one all_gather_object; one entry without a process group).  bench.py prints it as `per_rank_device` next to
    `ranks_seen`, so that whoever runs the 8-GPU line first can see that eight ranks were on eight GPUs."""
    import socket

    import torch
    import torch.distributed as dist

    me = {"rank": int(os.environ.get("RANK", "0")), "host": socket.gethostname(), "device": None, "pci_bus_id": None, "name": None}
    if device is not None and torch.cuda.is_available():
        idx = device.index if hasattr(device, "index") and device.index is not None else torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        me["device"] = int(idx)
        me["name"] = props.name
        bus = getattr(props, "pci_bus_id", None)
        me["pci_bus_id"] = (f"{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(props, 'pci_device_id', 0):02x}"
                            if bus is not None else (str(getattr(props, "uuid", "")) or None))
    if not _collectives_on():
        return [me]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out


def ranks_sharing_a_device(ranks):
    """[(rank, rank)] pairs of `describe_ranks()` entries that sit on the same GPU of the same host."""
    seen, clashes = {}, []
    for r in ranks:
        key = (r["host"], r["pci_bus_id"] if r["pci_bus_id"] is not None else r["device"])
        if key in seen:
            clashes.append((seen[key], r["rank"]))
        else:
            seen[key] = r["rank"]
    return clashes
