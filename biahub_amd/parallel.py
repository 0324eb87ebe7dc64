"""Sharding positions over the GPUs of one node — one process per GPU, no data-path collective.

The reference fans positions out as independent Slurm/submitit jobs (biahub/deskew.py:715-749); every
(position, t, c) volume is independent.  Here rank r takes positions r, r+W, r+2W, ... and the only exchange is a
barrier plus a tiny all_gather of per-rank status before rank 0 finalises plate-level metadata (RCCL over xGMI on
GPUs; the same code runs over gloo on CPU, which is how tests cover it).
"""

from __future__ import annotations

import os
import time
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class RankStats:
    n_done: int = 0
    n_failed: int = 0
    seconds: float = 0.0
    voxels: float = 0.0


def world_info():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_positions(positions, rank: int, world: int):
    """Round-robin shard: position i belongs to rank i % world."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(positions)[rank::world]


def init(backend: str | None = None, device: torch.device | None = None):
    """Initialise torch.distributed when WORLD_SIZE > 1 ('nccl' == RCCL on GPUs, 'gloo' on CPU)."""
    rank, _, world = world_info()
    if world > 1 and not dist.is_initialized():
        # BH_DIST_BACKEND=gloo lets a multi-rank run be rehearsed on a box with fewer GPUs than ranks
        backend = os.environ.get("BH_DIST_BACKEND") or backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def barrier(device=None):
    if dist.is_initialized():
        dist.barrier()


def _comm_device(device):
    """Collectives run on the GPU for RCCL and on the host for gloo."""
    if device is None or (dist.is_initialized() and dist.get_backend() == "gloo"):
        return "cpu"
    return device


def max_over_ranks(value: float, device=None) -> float:
    """The slowest rank's value (bench.py's timing contract)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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


def process_positions(positions, func, rank: int, world: int) -> RankStats:
    """Run ``func(position)`` (returns voxels processed) on this rank's shard; failures are counted, not fatal,
    so one bad position does not take the plate down (the reference isolates positions as separate jobs)."""
    st = RankStats()
    t0 = time.perf_counter()
    for pos in shard_positions(positions, rank, world):
        try:
            st.voxels += float(func(pos))
            st.n_done += 1
        except Exception:  # noqa: BLE001 - counted and reported by rank 0
            st.n_failed += 1
    st.seconds = time.perf_counter() - t0
    return st
