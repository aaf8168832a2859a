#!/usr/bin/env python
"""GPU box: C3 rollout rate with the per-step output delivered to the HOST (the PCIe-inclusive rate): the pinned ring of
wxengine.output against the reference's blocking `.cpu().numpy()` per step, and against no transfer at all."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT]
import torch  # noqa: E402

from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.output import rollout_to_host  # noqa: E402
from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict  # noqa: E402

cfg = named_config("C3")
eng = WXEngine(cfg, "bf16")
eng.load_state_dict(synth_state_dict(cfg))
eng.finalize()
eng.set_denorm(*synth_denorm(cfg.base_output_channels))
n_prog = cfg.channels * cfg.levels + cfg.surface_channels
eng.set_layout(n_prog, cfg.base_input_channels - n_prog - 2, 2)
x0 = torch.from_numpy(synth_input(cfg)).cuda()
n = 40
frc = [torch.from_numpy(synth_forcing(cfg, 2, t % 4)).cuda() for t in range(4)]
frcs = [frc[t % 4] for t in range(n)]


def timed(fn):
    fn(3)
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t)


def no_transfer(k):
    x = x0
    for t in range(k):
        _y, _yp, x = eng.step(x, frcs[t], want_y=False)


def blocking(k):   # rollout_to_netcdf.py:292
    x = x0
    for t in range(k):
        _y, yp, x = eng.step(x, frcs[t], want_y=False)
        yp.cpu().numpy()


sink = []


def ring(k):
    rollout_to_host(eng, x0, frcs[:k], lambda i, a: sink.append(float(a[0, 0, 0, 0])))


mb = cfg.base_output_channels * cfg.out_hw[0] * cfg.out_hw[1] * 4 / 1e6
print(f"C3 bf16, {mb:.0f} MB of output per step")
print(f"  no transfer (bench.py metric)        {timed(no_transfer):7.2f} steps/s")
print(f"  pinned double-buffered ring           {timed(ring):7.2f} steps/s   (PCIe-inclusive)")
print(f"  blocking .cpu().numpy() per step      {timed(blocking):7.2f} steps/s   (what the reference loop does)")
