#!/usr/bin/env python
"""Dev helper (GPU box): race screen of the two-stream half-map schedule -- N forwards and N 3-step rollouts of a two-stream engine against
a one-stream engine of the same weights, counting outputs that are not bit-identical.   python tools/two_stream_stress.py [C3] [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch  # noqa: E402

from ab_time import make  # noqa: E402
from wxengine.config import named_config  # noqa: E402
from wxengine.synth import synth_forcing, synth_input, synth_state_dict  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = named_config(name)
sd = synth_state_dict(cfg)
force = {} if name == "C3" else {"WX_STREAM_MIN_ROWS": "0"}
one, n_dyn = make(cfg, sd, "bf16", {**force, "WX_TWO_STREAM": "0"})
two, _ = make(cfg, sd, "bf16", {**force, "WX_TWO_STREAM": "1"})
x0 = torch.from_numpy(synth_input(cfg, seed=1000)).cuda()
frcs = [torch.from_numpy(synth_forcing(cfg, n_dyn, t, seed=1000)).cuda() for t in range(3)]
oh, ow = cfg.out_hw
bad_f = bad_r = 0
yref = one.forward(x0).clone()
torch.cuda.synchronize()
for i in range(n):
    y = two.forward(x0)
    torch.cuda.synchronize()
    if not torch.equal(y, yref):
        bad_f += 1
        print(f"forward {i}: max diff {float((y - yref).abs().max()):.3e}, {int((y != yref).sum())} elements", flush=True)
yp = torch.empty((1, cfg.base_output_channels, oh, ow), dtype=torch.float32, device="cuda")
xr = torch.empty_like(x0)
one.rollout(x0, frcs, [yp] * 3, x_final=xr)
torch.cuda.synchronize()
ref = (xr.clone(), yp.clone())
for i in range(n):
    xf = torch.empty_like(x0)
    two.rollout(x0, frcs, [yp] * 3, x_final=xf)
    torch.cuda.synchronize()
    if not (torch.equal(xf, ref[0]) and torch.equal(yp, ref[1])):
        bad_r += 1
        print(f"rollout {i}: max diff {float((yp - ref[1]).abs().max()):.3e}", flush=True)
    # and the one-stream engine against itself (is the reference deterministic?)
    one.rollout(x0, frcs, [yp] * 3, x_final=xf)
    torch.cuda.synchronize()
    if not (torch.equal(xf, ref[0]) and torch.equal(yp, ref[1])):
        print(f"ONE-STREAM rollout {i} differs from its own first run: max diff {float((yp - ref[1]).abs().max()):.3e}", flush=True)
print(f"[stress] {name}: {bad_f} of {n} forwards and {bad_r} of {n} rollouts of the two-stream engine differ from the one-stream engine")
