import os, sys
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT]
import numpy as np, torch
from wxengine.config import named_config
from wxengine.engine import WXEngine
from wxengine.synth import synth_input, synth_state_dict
GOLD = os.path.join(ROOT, "tests", "golden")
for name, fam in [("T0","base"),("T1","base"),("T0","stress"),("T1","stress"),("T0","stress_hi"),("C1","base"),("C1","stress"),("C1","stress_hi"),("C3S","base"),("C3","base")]:
    cfg = named_config(name)
    f = f"model_{name}.npz" if fam == "base" else f"model_{name}_{fam}.npz"
    g = np.load(os.path.join(GOLD, f)); s = int(g["stride"])
    for prec in ("fp32", "fp32s"):
        eng = WXEngine(cfg, prec, 0); eng.load_state_dict(synth_state_dict(cfg, family=fam)); eng.finalize()
        y = eng.forward(torch.from_numpy(synth_input(cfg)).cuda()).cpu()
        ys = y[0, :, 0, ::s, ::s].numpy().astype(np.float64)
        err = np.abs(ys - g["y"]).max() / np.abs(g["y"]).max()
        l2 = np.linalg.norm(ys - g["y"]) / np.linalg.norm(g["y"])
        print(f"{name:4s} {fam:10s} {prec:6s} max {err:.3e} relL2 {l2:.3e} split_gemms {eng.query('split_gemms')}", flush=True)
        del eng
