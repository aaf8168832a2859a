#!/usr/bin/env python
"""GPU box: the split-bf16 mode ("fp32s": fp32 storage, three bf16 MFMAs per product) beside the exact-f32 engine against the
reference's own fp32 forward (tests/golden/model_*.npz), base and stress weight families, small maps to the headline model.
Prints max|y - ref| / max|ref| (the stated tolerance is 1e-4) and rel-L2 per (config, family, precision)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.synth import synth_input, synth_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [("T0", "base"), ("T1", "base"), ("T0", "stress"), ("T1", "stress"), ("T0", "stress_hi"), ("C1", "base"), ("C1", "stress"),
         ("C1", "stress_hi"), ("C3S", "base"), ("C3S", "stress"), ("C3S", "stress_hi"), ("C3", "base"), ("C3", "stress"), ("C3", "stress_hi")]
for name, fam in CASES:
    cfg = named_config(name)
    f = f"model_{name}.npz" if fam == "base" else f"model_{name}_{fam}.npz"
    if not os.path.isfile(os.path.join(GOLD, f)):
        continue
    g = np.load(os.path.join(GOLD, f))
    s = int(g["stride"])
    for prec in ("fp32", "fp32s"):
        eng = WXEngine(cfg, prec, 0)
        eng.load_state_dict(synth_state_dict(cfg, family=fam))
        eng.finalize()
        y = eng.forward(torch.from_numpy(synth_input(cfg)).cuda()).cpu()
        ys = y[0, :, 0, ::s, ::s].numpy().astype(np.float64)
        err = np.abs(ys - g["y"]).max() / np.abs(g["y"]).max()
        l2 = np.linalg.norm(ys - g["y"]) / np.linalg.norm(g["y"])
        print(f"{name:4s} {fam:10s} {prec:6s} max {err:.3e} relL2 {l2:.3e} split_gemms {eng.query('split_gemms')}", flush=True)
        del eng
