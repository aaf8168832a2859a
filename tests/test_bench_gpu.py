"""bench.py end to end on the GPU box (tiny workload): the contract line, the self-launcher for --gpus N, the PCIe-inclusive object."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--config", "T1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-fp32", "--no-config2", "--no-concurrent"]


def _bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_one_gpu_line_carries_roofline_and_host_delivery():
    r, lines = _bench(QUICK)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["config"]["finite_outputs"]
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["achieved"] > 0
    hd = out["host_delivery"]   # every step's output delivered to pinned host memory (rollout_to_netcdf.py:289-301)
    assert hd["value"] > 0 and hd["d2h_GBps"] > 0 and hd["finite_outputs"] and hd["blocking_copy_per_step"]["value"] > 0


def test_gpus_2_without_torchrun_runs_two_ranks():
    """RANK unset: bench.py launches its own two ranks (gloo: they share the one GPU of this box -- a functional check of the N > 1
    path through the real engine; timings mean nothing).  The line must say n_gpus = 2 and count both ranks' steps."""
    r, lines = _bench(["--gpus", "2", "--no-roofline", "--no-host-delivery", *QUICK], WX_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["total_steps"] == 6 and out["scaling"] == "weak"
    assert "x2" in out["config"]["parallelism"]


def test_gpus_beyond_the_node_fail_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    r, lines = _bench(["--gpus", str(n), *QUICK])
    assert r.returncode != 0 and not lines and f"--gpus {n}" in r.stderr


def test_latband_two_ranks_through_the_self_launcher():
    r, lines = _bench(["--gpus", "2", "--latband", *QUICK], WX_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["finite_outputs"]


def test_forecast_pool_is_bit_identical_to_sequential_rollouts():
    """wxengine.replicas.ForecastPool: two forecasts in flight on one GPU (own engine + stream each) give the bits of running them one
    after the other -- the engines share nothing but the device."""
    import torch
    from wxengine.config import named_config
    from wxengine.engine import WXEngine
    from wxengine.replicas import ForecastPool
    from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict
    cfg = named_config("T1")
    sd = synth_state_dict(cfg)

    def make():
        e = WXEngine(cfg, "bf16", 0)
        e.load_state_dict(sd)
        e.finalize()
        e.set_denorm(*synth_denorm(cfg.base_output_channels))
        e.set_layout(cfg.channels * cfg.levels + cfg.surface_channels, 2, 2)
        return e
    pool = ForecastPool(make, 2)
    n = 4
    frc = [torch.from_numpy(synth_forcing(cfg, 2, t + 1)).cuda() for t in range(n)]
    shape = (1, cfg.base_output_channels) + tuple(cfg.out_hw)
    jobs, want = [], []
    for i in range(2):
        x0 = torch.from_numpy(synth_input(cfg, seed=1000 + i)).cuda()
        outs = [torch.empty(shape, device="cuda") for _ in range(n)]
        jobs.append(dict(x0=x0, forcings=frc, phys_out=outs, x_final=torch.empty_like(x0)))
        ref_out = [torch.empty(shape, device="cuda") for _ in range(n)]
        xf = torch.empty_like(x0)
        pool.engines[i].rollout(x0, frc, ref_out, x_final=xf)
        want.append((ref_out, xf))
    torch.cuda.synchronize()
    for _ in range(3):
        pool.rollout_all(jobs)
        torch.cuda.synchronize()
        for i in range(2):
            assert all(torch.equal(a, b) for a, b in zip(jobs[i]["phys_out"], want[i][0])) and torch.equal(jobs[i]["x_final"], want[i][1])
    assert not torch.equal(want[0][0][-1], want[1][0][-1])
    with pytest.raises(ValueError):
        pool.rollout_all(jobs + jobs)
