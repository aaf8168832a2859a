#!/usr/bin/env python
"""Dev helper (GPU box): layer-by-layer parity of the HIP engine against the oracle."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import wxformer_oracle as O  # noqa: E402
from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.synth import synth_input, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="T0")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--layers", action="store_true")
    args = ap.parse_args()
    cfg = named_config(args.config)
    sd = synth_state_dict(cfg)
    x = synth_input(cfg)
    cap = {}
    t = time.time()
    y_ref = O.forward(cfg, sd, x, capture=cap if args.layers else None)
    print(f"oracle forward {time.time() - t:.2f}s")
    eng = WXEngine(cfg, args.precision)
    eng.load_state_dict(sd)
    t = time.time()
    eng.finalize()
    print(f"finalize {time.time() - t:.2f}s")
    eng.set_debug(args.layers)
    xd = torch.from_numpy(x).cuda()
    y = eng.forward(xd)
    torch.cuda.synchronize()
    y = y.cpu()
    if args.layers:
        for k, v in cap.items():
            try:
                got = eng.debug_read(k)
            except Exception as e:
                print(f"  {k:34s} (no capture: {e})")
                continue
            ref = v[0].numpy()
            err = np.abs(got - ref).max()
            print(f"  {k:34s} max|ref|={np.abs(ref).max():9.4f} max err={err:10.3e} rel={err / max(np.abs(ref).max(), 1e-9):9.2e}")
    err = (y - y_ref).abs().max().item()
    rel_l2 = ((y - y_ref).norm() / y_ref.norm()).item()
    print(f"[{args.config} {args.precision}] max|y|={y_ref.abs().max():.4f} max err={err:.3e} "
          f"rel(max)={err / y_ref.abs().max().item():.3e} rel-L2={rel_l2:.3e} nan={torch.isnan(y).any().item()}")


if __name__ == "__main__":
    main()
