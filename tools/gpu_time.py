#!/usr/bin/env python
"""Dev helper (GPU box): time the engine on a named config and print the per-kernel-class profile."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.synth import synth_input, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--golden", action="store_true")
    ap.add_argument("--detail", action="store_true")
    args = ap.parse_args()
    cfg = named_config(args.config)
    t = time.time()
    sd = synth_state_dict(cfg)
    print(f"synth weights {time.time() - t:.1f}s")
    eng = WXEngine(cfg, args.precision)
    eng.load_state_dict(sd)
    t = time.time()
    eng.finalize()
    print(f"finalize {time.time() - t:.1f}s")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    y = eng.forward(x)
    torch.cuda.synchronize()
    if args.golden:
        g = np.load(os.path.join(ROOT, "tests", "golden", f"model_{args.config}.npz"))
        s = int(g["stride"])
        ys = y[0, :, 0, ::s, ::s].cpu().numpy()
        err = np.abs(ys - g["y"]).max()
        rl2 = np.linalg.norm(ys - g["y"]) / np.linalg.norm(g["y"])
        print(f"[{args.config} {args.precision}] vs reference golden: max err {err:.3e} (max|y| {np.abs(g['y']).max():.3f}) rel-L2 {rl2:.3e}")
    for _ in range(2):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(args.steps):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    dt = (time.time() - t) / args.steps
    print(f"[{args.config} {args.precision}] {dt * 1e3:.2f} ms/forward")
    eng.profile(2 if args.detail else 1)
    eng.profile_reset()
    eng.forward(x, out=y)
    rows = eng.profile_read()
    tot = sum(r["ms"] for r in rows)
    for r in sorted(rows, key=lambda r: -r["ms"]):
        tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
        gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0
        print(f"  {r['name']:18s} n={r['launches']:4d} {r['ms']:9.3f} ms {100 * r['ms'] / tot:5.1f}%  {tf:8.1f} TF/s {gb:8.0f} GB/s")
    print(f"  total (event sum) {tot:.2f} ms")


if __name__ == "__main__":
    main()
