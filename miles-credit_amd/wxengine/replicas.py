"""Multi-GPU mode "replicas over init times" (SURVEY.md §8(e) mode 1): the reference's own inference parallelism
(credit/applications/rollout_to_netcdf.py:259 -- forecast i runs on rank i % world_size).  One process per GPU, every rank
advances its own forecast(s); there is NO collective on the data path.  What the ranks share is the clock: `bench.py` (and
any rollout driver) brackets the timed region with a barrier on both sides and reduces the elapsed time with MAX.

`ReplicaGroup` is that harness.  `bench.py --gpus N` runs through it with the RCCL backend ("nccl"); the world-size-2 CPU test
(tests/test_dist_cpu.py) runs the same object over gloo."""
from __future__ import annotations

import os
import time
from typing import Callable, List, Optional, Sequence


class ReplicaGroup:
    def __init__(self, backend: str = "nccl", n_expected: Optional[int] = None, device_index: Optional[int] = None):
        """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun sets them.  `device_index`: the GPU of this rank for the
        nccl backend (default LOCAL_RANK); with another backend ranks may share devices (functional checks only)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if n_expected is not None and self.world > 1 and self.world != n_expected:
            raise SystemExit(f"--gpus {n_expected} but WORLD_SIZE={self.world}")
        self.backend = backend
        self.dist = None
        self._reduce_device = "cpu"
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dev = torch.device("cuda", self.local_rank if device_index is None else device_index)
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=dev)
                self._reduce_device = dev
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    # ---- work split -----------------------------------------------------------------------------------
    def my_share(self, forecasts: Sequence) -> List:
        """rollout_to_netcdf.py:259: forecast i belongs to rank i % world_size."""
        return [f for i, f in enumerate(forecasts) if i % self.world == self.rank]

    # ---- the shared clock ---------------------------------------------------------------------------------
    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self._reduce_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_true(self, flag: bool) -> bool:
        return self.max_over_ranks(0.0 if flag else 1.0) == 0.0

    def timed(self, work: Callable[[], None], device_sync: Callable[[], None]) -> float:
        """Seconds for `work()` on the SLOWEST rank: barrier + device sync on both sides of the region, MAX over ranks."""
        self.barrier()
        device_sync()
        t0 = time.perf_counter()
        work()
        device_sync()
        self.barrier()
        device_sync()
        return self.max_over_ranks(time.perf_counter() - t0)

    def throughput(self, steps_per_rank: int, elapsed_s: float) -> float:
        """Whole-job forecast-steps/sec: every rank advanced `steps_per_rank` steps of its own forecast."""
        return steps_per_rank * self.world / elapsed_s

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
