"""GPU box: sharded (virtual ranks on one GPU) vs unsharded engine.   python tools/band_check.py T1 fp32 2 [3 ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "miles-credit_amd"))
from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.latband import VirtualBands  # noqa: E402
from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict  # noqa: E402

name, prec = sys.argv[1], sys.argv[2]
cfg = named_config(name)
sd = synth_state_dict(cfg)
x = torch.from_numpy(synth_input(cfg)).cuda()
n_prog = cfg.channels * cfg.levels + cfg.surface_channels
n_dyn = min(2, cfg.base_input_channels - n_prog)
n_static = cfg.base_input_channels - n_prog - n_dyn
frc = torch.from_numpy(synth_forcing(cfg, n_dyn, 1)).cuda() if n_dyn else None
mean, std = synth_denorm(cfg.base_output_channels)


def setup(e):
    e.set_denorm(mean, std)
    e.set_layout(n_prog, n_static, n_dyn)


ref = WXEngine(cfg, prec, 0)
ref.load_state_dict(sd)
ref.finalize()
setup(ref)
y0, p0, n0 = ref.step(x, frc)
prev = None
for n in [int(a) for a in sys.argv[3:]]:
    vb = VirtualBands(cfg, sd, n, prec, setup=setup)
    y, p, xn = vb.step(x, frc, want_phys=True, want_next=True)
    torch.cuda.synchronize()
    sc = y0.abs().max().item()
    print(f"{name} {prec} n={n}: rows {vb.starts}  max|y-y0|/max|y0| = {(y - y0).abs().max().item() / sc:.3e}  "
          f"phys {(p - p0).abs().max().item() / p0.abs().max().item():.3e}  x_next {(xn - n0).abs().max().item():.3e}  "
          f"exchanged {vb.exchanged_bytes / 1e6:.2f} MB")
    if prev is not None:
        print(f"   vs the previous sharding: max|dy| = {(y - prev).abs().max().item():.3e}")
    prev = y
    if len(sys.argv) > 3 and os.environ.get("BAND_ROWS"):
        d = (y - y0).abs().amax(dim=(0, 1, 2, 4)).cpu().numpy()
        print("  per-row max err:", np.array2string(d, precision=1, max_line_width=200))
