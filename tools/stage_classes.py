"""GPU box: per kernel class and stage, ms per step and launches (profile mode 2).   python tools/stage_classes.py C3 bf16 [filter]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "miles-credit_amd"))
from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.synth import synth_input, synth_state_dict  # noqa: E402

name, prec = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
cfg = named_config(name)
eng = WXEngine(cfg, prec, 0)
eng.load_state_dict(synth_state_dict(cfg))
eng.finalize()
x = torch.from_numpy(synth_input(cfg)).cuda()
eng.profile(2)
for _ in range(2):
    eng.step(x, None, want_phys=False, want_next=False)
eng.profile_reset()
K = 5
for _ in range(K):
    eng.step(x, None, want_phys=False, want_next=False)
torch.cuda.synchronize()
rows = sorted(eng.profile_read(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rows) / K
print(f"{name} {prec}: {tot:.3f} ms kernel time / step")
for r in rows:
    if flt in r["name"]:
        tf = r["flops"] / (r["ms"] * 1e-3) * 1e-12 if r["ms"] > 0 else 0.0     # algorithmic FLOP / measured time
        tb = r["bytes"] / (r["ms"] * 1e-3) * 1e-12 if r["ms"] > 0 else 0.0     # algorithmic bytes / measured time
        print(f"  {r['name']:<26s} {r['ms'] / K:8.3f} ms  {r['launches'] // K:4d} launches  {1e3 * r['ms'] / r['launches']:8.1f} us each"
              f"  {tf:7.0f} TFLOP/s  {tb:5.2f} TB/s")
