"""One-process-per-GPU helpers for the data-parallel hot path (bench.py, tests).

The raster path shards on the scene axis with no exchange step, so the only
collectives here are the barrier around the timed region and the MAX over ranks
of the elapsed time (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in
the CPU tests)."""
from __future__ import annotations

import os
import time
from typing import Callable, Optional

import torch


def env_world() -> tuple:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def launched_by_torchrun() -> bool:
    """True inside a `python -m torch.distributed.run` worker (it exports the rendezvous of the group it expects us to join)."""
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))


def init_distributed(backend: str, device: Optional[torch.device] = None, single_rank_group: bool = False):
    """Returns the torch.distributed module if WORLD_SIZE > 1 (process group initialised), else None.  `single_rank_group`: also
    initialise a ONE-rank group (a `torch.distributed.run --nproc-per-node 1` launch: the RCCL path then runs on one GPU)."""
    rank, _, world = env_world()
    if world <= 1 and not single_rank_group:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def scene_seeds(rank: int, scenes_per_rank: int, base: int = 1234) -> list:
    """Disjoint scene shards: rank r renders scenes base + 1000 r + i (per-rank seeding as main_style.py:118)."""
    assert scenes_per_rank < 1000
    return [base + 1000 * rank + i for i in range(scenes_per_rank)]


def timed_steps(step: Callable[[], object], steps: int, sync: Callable[[], None], dist=None,
                device: Optional[torch.device] = None) -> float:
    """barrier + sync, exactly `steps` calls, sync + barrier; returns the MAX elapsed seconds over ranks."""
    def fence():
        sync()
        if dist is not None:
            dist.barrier()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_throughput(units_per_rank_per_step: int, steps: int, world: int, max_seconds: float) -> float:
    """whole-job units/s: all ranks' units over the slowest rank's time (weak scaling)."""
    return units_per_rank_per_step * world * steps / max_seconds
