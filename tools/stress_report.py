#!/usr/bin/env python
"""GPU box: error of the engine on the stress weight families (wxengine.synth.FAMILIES) against the reference goldens, per config /
precision, and layer by layer against the oracle (where the error enters).  `python tools/stress_report.py [family ...]`"""
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


def err(y, ref):
    y, ref = np.asarray(y, np.float64), np.asarray(ref, np.float64)
    return np.linalg.norm(y - ref) / max(np.linalg.norm(ref), 1e-300), np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-300)


def main():
    fams = sys.argv[1:] or ["base", "stress", "stress_hi"]
    layers = os.environ.get("WX_REPORT_LAYERS", "0") == "1"
    for fam in fams:
        for name in ("T0", "T1", "C1"):
            cfg = named_config(name)
            sd = synth_state_dict(cfg, family=fam)
            x = synth_input(cfg)
            g = np.load(os.path.join(GOLD, f"model_{name}.npz" if fam == "base" else f"model_{name}_{fam}.npz"))
            s = int(g["stride"])
            cap = None
            if layers and name == "T0":
                from oracle import wxformer_oracle as O
                cap = {}
                O.forward(cfg, sd, x, capture=cap)
            for prec in ("fp32", "bf16"):
                eng = WXEngine(cfg, prec, 0)
                eng.load_state_dict(sd)
                eng.finalize()
                if cap is not None:
                    eng.set_debug(True)
                y = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
                l2, mx = err(y[0, :, 0, ::s, ::s], g["y"])
                print(f"{fam:10s} {name:3s} {prec}: rel-L2 {l2:.3e}  max/scale {mx:.3e}  finite {bool(np.isfinite(y).all())}", flush=True)
                if cap is not None:
                    for k, v in cap.items():
                        try:
                            got = eng.debug_read(k)
                        except Exception:
                            continue
                        ref = v[0].numpy()
                        l2k, mxk = err(got, ref)
                        # the same with every pixel's channel mean removed: what a LayerNorm behind this map sees
                        l2c, _ = err(got - got.mean(axis=0, keepdims=True), ref - ref.mean(axis=0, keepdims=True))
                        r = np.abs(ref.mean(axis=0)) / np.maximum(ref.std(axis=0), 1e-30)
                        print(f"      {k:36s} rel-L2 {l2k:.3e} max/scale {mxk:.3e}  centred rel-L2 {l2c:.3e}  |mean|/sigma median {np.median(r):.1f}")
                del eng


if __name__ == "__main__":
    main()
