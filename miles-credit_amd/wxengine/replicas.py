"""Multi-GPU mode 'replicas over init times' (SURVEY.md §8(e)): the reference's own inference
parallelism (credit/applications/rollout_to_netcdf.py:259, `forecasts[i] -> rank i % world`).
No collective in the step loop; one barrier + a max-reduce of the wall time at the end."""
from __future__ import annotations

from typing import List, Sequence


def shard_init_times(all_forecasts: Sequence, rank: int, world_size: int) -> List:
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return [f for i, f in enumerate(all_forecasts) if i % world_size == rank]


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """All-reduce(MAX) of a python float; identity without a process group."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(steps_per_rank: int, world_size: int, max_elapsed_s: float) -> float:
    """Whole-job forecast-steps/sec: every rank advanced `steps_per_rank` steps of its own forecast."""
    return steps_per_rank * world_size / max_elapsed_s
