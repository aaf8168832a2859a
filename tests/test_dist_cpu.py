"""world_size-2 gloo test of the N>1 path: the SAME harness object `bench.py --gpus N` times its rollout with
(wxengine.replicas.ReplicaGroup: init-time sharding, barrier-bracketed region, MAX-over-ranks clock, aggregate throughput)."""
import json
import os
import socket
import subprocess
import sys
import time

import torch.multiprocessing as mp

from wxengine.replicas import ReplicaGroup


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    grp = ReplicaGroup(backend="gloo", n_expected=world)
    mine = grp.my_share(list(range(7)))
    steps = []

    def work():                      # rank 1 is the slow one: the shared clock must report ITS time on every rank
        for t in range(4):
            time.sleep(0.05 * (1 + rank))
            steps.append(t)
    elapsed = grp.timed(work, lambda: None)
    gathered = [None] * world
    grp.dist.all_gather_object(gathered, mine)
    q.put((rank, mine, elapsed, gathered, grp.throughput(len(steps), elapsed), grp.all_true(rank == 0), grp.all_true(True)))
    grp.close()


def test_replica_group_two_ranks_gloo():
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
    (_, m0, e0, g0, v0, f0, t0), (_, m1, e1, g1, v1, f1, t1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]          # rollout_to_netcdf.py:259 rule
    assert sorted(g0[0] + g0[1]) == list(range(7)) and g0 == g1
    assert e0 == e1 and e0 >= 4 * 0.05 * 2                  # MAX over ranks: both see the slow rank's time
    assert v0 == v1 and abs(v0 - 2 * 4 / e0) < 1e-9         # whole-job steps/s = world * steps / max time
    assert (f0, f1) == (False, False) and (t0, t1) == (True, True)


def test_single_process_identity(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    grp = ReplicaGroup(backend="gloo")
    assert grp.world == 1 and grp.dist is None
    assert grp.my_share(["a", "b", "c"]) == ["a", "b", "c"]
    assert grp.max_over_ranks(3.5) == 3.5
    calls = []
    e = grp.timed(lambda: calls.append("work"), lambda: calls.append("sync"))
    assert calls == ["sync", "work", "sync", "sync"] and e >= 0
    assert grp.throughput(40, 2.0) == 20.0


_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miles-credit_amd")


def _run_selftest(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(PYTHONPATH=_PKG + os.pathsep + e.get("PYTHONPATH", ""), **env)
    return subprocess.run([sys.executable, "-m", "wxengine.replicas", *args], env=e, capture_output=True, text=True, timeout=180)


def test_gpus_n_without_a_launcher_spawns_n_ranks():
    """`--gpus 2` with RANK unset: the process becomes the launcher (wxengine.replicas.ensure_ranks, the first thing bench.py calls),
    two gloo ranks run the harness, rank 0 prints ONE line with n_gpus = 2 -- not a silent single-rank run (missing #3)."""
    r = _run_selftest(["--gpus", "2", "--steps", "3"], WX_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["mine"] == [0, 2, 4] and out["value"] > 0


def test_gpus_n_on_a_node_with_fewer_gpus_fails_loudly():
    r = _run_selftest(["--gpus", "64"], WX_BENCH_BACKEND="nccl")   # no node has 64 GPUs
    assert r.returncode != 0 and "--gpus 64" in r.stderr and "GPU(s)" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_under_an_external_launcher_is_rejected():
    r = _run_selftest(["--gpus", "4"], WX_BENCH_BACKEND="gloo", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_PORT=str(_free_port()))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_a_failing_rank_takes_the_job_down():
    from wxengine.replicas import launch_ranks
    t0 = time.time()
    rc = launch_ranks(2, ["-c", "import os, sys, time; sys.exit(3) if os.environ['RANK'] == '1' else time.sleep(60)"])
    assert rc == 3 and time.time() - t0 < 30
