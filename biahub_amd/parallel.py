"""Sharding positions over the GPUs of one node — one process per GPU, no data-path collective.

The reference fans positions out as independent Slurm/submitit jobs (biahub/deskew.py:715-749); every
(position, t, c) volume is independent.  Here rank r takes positions r, r+W, r+2W, ... and the only exchange is a
barrier plus a tiny all_gather of per-rank status before rank 0 finalises plate-level metadata (RCCL over xGMI on
GPUs; the same code runs over gloo on CPU, which is how tests cover it).

``init()`` is the one place a rank is bound to its GPU: ``LOCAL_RANK % device_count`` becomes torch's current device
*before* anything touches a GPU, so every later ``device="cuda"`` (what the reference's operators are handed,
biahub/deskew.py:712) resolves to this rank's card, and the collectives below run on it.
"""

from __future__ import annotations

import os
import sys
import time
import traceback
from dataclasses import dataclass

import torch
import torch.distributed as dist

_DEVICE: torch.device | None = None  # the GPU init() bound this rank to (None: no GPU / not bound)


@dataclass
class RankStats:
    n_done: int = 0
    n_failed: int = 0
    seconds: float = 0.0
    voxels: float = 0.0
    first_error: str = ""  # "<position>: <exception>" of this rank's first failure (local; not gathered)


def world_info():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_positions(positions, rank: int, world: int):
    """Round-robin shard: position i belongs to rank i % world."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(positions)[rank::world]


def bind_device() -> torch.device | None:
    """Make GPU ``LOCAL_RANK % device_count`` this process's current device and remember it.

    ``torch.cuda.device_count()`` does not initialise a GPU, so this is safe to call first thing in a command.  More
    ranks than GPUs (a rehearsal with BH_DIST_BACKEND=gloo) wrap around.  Returns None on a box without GPUs.
    """
    global _DEVICE
    n = torch.cuda.device_count()
    if n < 1:
        _DEVICE = None
        return None
    _, local_rank, _ = world_info()
    _DEVICE = torch.device("cuda", local_rank % n)
    torch.cuda.set_device(_DEVICE)
    return _DEVICE


def bound_device() -> torch.device | None:
    """The GPU ``init()`` / ``bind_device()`` gave this rank."""
    return _DEVICE


def init(backend: str | None = None, device: torch.device | None = None):
    """Bind this rank to its GPU, then initialise torch.distributed when WORLD_SIZE > 1 ('nccl' == RCCL on GPUs,
    'gloo' on CPU).  ``device`` overrides the LOCAL_RANK binding (bench.py passes the one it already set)."""
    global _DEVICE
    rank, _, world = world_info()
    if device is not None:
        _DEVICE = torch.device(device)
        if _DEVICE.type == "cuda":
            torch.cuda.set_device(_DEVICE)
    elif _DEVICE is None:
        bind_device()
    if world > 1 and not dist.is_initialized():
        # BH_DIST_BACKEND=gloo lets a multi-rank run be rehearsed on a box with fewer GPUs than ranks
        backend = os.environ.get("BH_DIST_BACKEND") or backend or ("nccl" if _DEVICE is not None else "gloo")
        kw = {"device_id": _DEVICE} if (backend == "nccl" and _DEVICE is not None and _DEVICE.type == "cuda") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def _comm_device(device):
    """Collectives run on this rank's GPU for RCCL and on the host for gloo."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return "cpu"
    device = device if device is not None else _DEVICE
    return "cpu" if device is None else device


def barrier(device=None):
    if dist.is_initialized():
        dev = _comm_device(device)
        if dev != "cpu":
            dist.barrier(device_ids=[torch.device(dev).index])
        else:
            dist.barrier()


class Rank0Error(RuntimeError):
    """Raised on ranks > 0 when the work rank 0 does for everybody (plate lay-out, transfer function) failed there."""


def rank0_first(fn, *args, **kwargs):
    """Run ``fn`` on rank 0 only, then let every rank learn whether it worked: rank 0 re-raises its own exception, the
    others raise :class:`Rank0Error` with its text.  Every rank reaches the collective on both paths — a plain
    ``if rank == 0: fn(); barrier()`` leaves the other ranks in the barrier until the collective times out (about ten
    minutes, GPUs held) when rank 0 raises before it gets there."""
    rank, _, _ = world_info()
    err: BaseException | None = None
    result = None
    if rank == 0:
        try:
            result = fn(*args, **kwargs)
        except BaseException as e:  # noqa: BLE001 - re-raised below, after the collective
            err = e
    if dist.is_initialized():
        box = [None if err is None else f"{type(err).__name__}: {err}"]
        dev = _comm_device(None)
        dist.broadcast_object_list(box, src=0, device=None if dev == "cpu" else torch.device(dev))
        if rank != 0 and box[0] is not None:
            raise Rank0Error(f"rank 0 failed: {box[0]}")
    if err is not None:
        raise err
    return result


def max_over_ranks(value: float, device=None) -> float:
    """The slowest rank's value (bench.py's timing contract)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj):
    """Every rank's (picklable) ``obj`` on every rank, in rank order — bench.py's evidence of which GPU each rank sits on
    (device name + PCI bus id over the job's own process group: RCCL at N > 1)."""
    if not dist.is_initialized():
        return [obj]
    rows = [None] * dist.get_world_size()
    dist.all_gather_object(rows, obj)
    return rows


def gather_stats(stats: RankStats, device=None):
    """all_gather of 4 doubles per rank; every rank gets the list (rank 0 writes the plate metadata)."""
    mine = torch.tensor([stats.n_done, stats.n_failed, stats.seconds, stats.voxels], dtype=torch.float64,
                        device=_comm_device(device))
    if not dist.is_initialized():
        rows = [mine]
    else:
        rows = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
    return [RankStats(int(r[0]), int(r[1]), float(r[2]), float(r[3])) for r in (x.cpu() for x in rows)]


def process_positions(positions, func, rank: int, world: int, label=str) -> RankStats:
    """Run ``func(position)`` (returns voxels processed) on this rank's shard.  A failing position is logged with its
    traceback to stderr and counted, not fatal to the rest of the shard: the reference isolates positions as separate
    jobs (biahub/deskew.py:733-749), so one bad position does not take the plate down; the caller turns
    ``n_failed > 0`` into a non-zero exit on the rank that saw it."""
    st = RankStats()
    t0 = time.perf_counter()
    for pos in shard_positions(positions, rank, world):
        try:
            st.voxels += float(func(pos))
            st.n_done += 1
        except Exception as e:  # noqa: BLE001 - logged here, counted, reported by the caller
            st.n_failed += 1
            if not st.first_error:
                st.first_error = f"{label(pos)}: {type(e).__name__}: {e}"
            print(f"[rank {rank}] position {label(pos)} failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
    st.seconds = time.perf_counter() - t0
    return st
