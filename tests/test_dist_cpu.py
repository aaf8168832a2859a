"""world_size-2 gloo test of the N>1 path (replicas over init times; no data-path collective)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wxengine.replicas import aggregate_throughput, max_over_ranks, shard_init_times


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_init_times(list(range(7)), rank, world)
    dist.barrier()
    elapsed = 1.0 + rank  # pretend rank 1 is slower
    mx = max_over_ranks(elapsed, dist)
    # ranks own disjoint forecasts whose union is everything
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, mine, mx, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sharding_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, mx0, g0), (r1, m1, mx1, g1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]          # rollout_to_netcdf.py:259 rule
    assert mx0 == mx1 == 2.0                               # MAX over ranks
    assert sorted(g0[0] + g0[1]) == list(range(7)) and g0 == g1
    assert aggregate_throughput(40, world, mx0) == 40.0


def test_single_process_identity():
    assert shard_init_times(["a", "b", "c"], 0, 1) == ["a", "b", "c"]
    assert max_over_ranks(3.5) == 3.5
