"""Multi-GPU mode "replicas over init times" (SURVEY.md §8(e) mode 1): the reference's own inference parallelism
(credit/applications/rollout_to_netcdf.py:259 -- forecast i runs on rank i % world_size).  One process per GPU, every rank
advances its own forecast(s); there is NO collective on the data path.  What the ranks share is the clock: `bench.py` (and
any rollout driver) brackets the timed region with a barrier on both sides and reduces the elapsed time with MAX.

`ReplicaGroup` is that harness.  `bench.py --gpus N` runs through it with the RCCL backend ("nccl"); the world-size-2 CPU test
(tests/test_dist_cpu.py) runs the same object over gloo."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Callable, List, Optional, Sequence


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv: Sequence[str], extra_env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Start `n` copies of `python argv...` on this node, one per rank, with the environment torchrun would give them (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT) -- what the reference gets from its launcher through
    credit/distributed.py:193-292 (rank discovery from the environment).  Rank 0 inherits stdout (the one JSON line of bench.py); if any
    rank fails the others are terminated (by PID) and the first non-zero exit code is returned."""
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n))
    env0.update(extra_env or {})
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, *argv], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc, t0 = 0, time.time()
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            if rc == 0:
                rc = 124
            break
        time.sleep(0.05)
    return rc


def ensure_ranks(n_gpus: int, backend: str, argv: Optional[Sequence[str]] = None) -> None:
    """`bench.py --gpus N` (or any driver) started WITHOUT a launcher: become the launcher.  With RANK already in the environment
    (torchrun, or a child of this function) it only checks WORLD_SIZE against N.  Fails loudly -- never silently runs one rank -- when
    the node has fewer GPUs than ranks (the RCCL backend needs one device per rank; WX_BENCH_BACKEND=gloo lets ranks share a device
    for functional checks)."""
    if "RANK" in os.environ:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks")
        return
    if n_gpus <= 1:
        return
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n_gpus:
            raise SystemExit(f"--gpus {n_gpus}: this node has {have} GPU(s); one rank per GPU is required over RCCL "
                             f"(set WX_BENCH_BACKEND=gloo for a functional check with shared devices)")
    raise SystemExit(launch_ranks(n_gpus, list(argv if argv is not None else sys.argv)))


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


class ForecastPool:
    """Several forecasts IN FLIGHT on one GPU: the reference gives every rank a list of init times and walks it one forecast at a time
    (rollout_to_netcdf.py:259-262); here a rank may advance `n` of its forecasts concurrently -- one engine (own activation buffers,
    shared nothing but the device) and one HIP stream per forecast in flight, one host thread each (the C ABI releases the GIL).  The
    small launches of the deep stages (one round of latency-bound workgroups) of one forecast then run beside the chip-filling kernels
    of the other: measured +7 % forecast-steps/s with n = 2 on MI355X (0.25 degree, bf16: 125.0 -> 134.0; n = 3: 131.3), at 1.87 x the
    per-step latency of a single forecast.  Results are bit-identical to running the forecasts one after the other."""

    def __init__(self, make_engine: Callable[[], object], n: int = 2, device: Optional[int] = None):
        import torch
        if n < 1:
            raise ValueError("ForecastPool: n >= 1")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.engines = [make_engine() for _ in range(n)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n)]

    def rollout_all(self, jobs: Sequence[dict]) -> None:
        """jobs[i] = kwargs of WXEngine.rollout (x0, forcings, phys_out, x_final) for forecast i; len(jobs) <= n.  Returns when every
        forecast has finished on the device."""
        import threading
        import torch
        if len(jobs) > len(self.engines):
            raise ValueError("ForecastPool: more jobs than engines")
        errors: List[BaseException] = []

        def work(i):
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(self.streams[i]):
                    self.engines[i].rollout(**jobs[i])
            except BaseException as e:  # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams[:len(jobs)]:
            st.wait_stream(cur)                     # inputs produced on the caller's stream are visible to the forecast streams
        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for st in self.streams[:len(jobs)]:
            cur.wait_stream(st)
        if errors:
            raise errors[0]


def _selftest(argv: Sequence[str]) -> None:
    """`python -m wxengine.replicas --gpus N [--steps K]`: the harness alone (no engine, no GPU) through the same entry sequence as
    bench.py -- ensure_ranks -> ReplicaGroup -> timed region -> rank 0 prints one JSON line.  tests/test_dist_cpu.py runs it."""
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args(list(argv))
    backend = os.environ.get("WX_BENCH_BACKEND", "nccl")
    ensure_ranks(args.gpus, backend, ["-m", "wxengine.replicas", *argv])
    grp = ReplicaGroup(backend=backend, n_expected=args.gpus)
    done = []

    def work():
        for t in range(args.steps):
            time.sleep(0.01 * (1 + grp.rank))
            done.append(t)
    elapsed = grp.timed(work, lambda: None)
    if grp.rank == 0:
        print(json.dumps({"n_gpus": grp.world, "steps": args.steps, "value": grp.throughput(len(done), elapsed),
                          "mine": grp.my_share(list(range(5)))}), flush=True)
    grp.close()


if __name__ == "__main__":
    _selftest(sys.argv[1:])
