"""GPU box: the FuXi-6h forward (BASELINE config 5) a few times -- run under rocprofv3 --kernel-trace --stats for the per-kernel table.
    python tools/fuxi_time.py [precision] [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "miles-credit_amd"))
from wxengine.fuxi import FuxiHIP, named_fuxi_config, synth_fuxi_state_dict  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = named_fuxi_config("F6H")
m = FuxiHIP(precision=prec, cfg=cfg)
m.load_state_dict(synth_fuxi_state_dict(cfg))
x = torch.randn(1, cfg.in_chans, cfg.frames, cfg.image_height, cfg.image_width, generator=torch.Generator().manual_seed(5)).cuda()
y = m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    m(x, y)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
print(f"FuXi-6h {prec}: {ms:.3f} ms per forward, {m.flops / ms / 1e9:.0f} TFLOP/s algorithmic, finite {bool(torch.isfinite(y).all())}")
