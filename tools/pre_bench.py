#!/usr/bin/env python
"""GPU box: HBM rate of the fused normalise + concatenate kernel (wx_pre_apply) at the C3 input size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from wxengine.preblock import DevicePreblock  # noqa: E402

H, W, L = 721, 1440, 13
fields = {"era5": {}}
for v in ("U", "V", "T", "Q"):
    fields["era5"][f"era5/prognostic/3d/{v}"] = torch.randn(1, L, 1, H, W, device="cuda")
for v in ("SP", "t2m", "V500", "U500"):
    fields["era5"][f"era5/prognostic/2d/{v}"] = torch.randn(1, 1, 1, H, W, device="cuda")
for v in ("Z_GDS4_SFC", "LSM"):
    fields["era5"][f"era5/static/2d/{v}"] = torch.randn(1, 1, 1, H, W, device="cuda")
for v in ("tsi", "sza"):
    fields["era5"][f"era5/dynamic_forcing/2d/{v}"] = torch.randn(1, 1, 1, H, W, device="cuda")
mean = {k.split("/")[-1]: np.full(t.shape[1], 0.5, np.float32) for k, t in fields["era5"].items()}
std = {k.split("/")[-1]: np.full(t.shape[1], 2.0, np.float32) for k, t in fields["era5"].items()}
pre = DevicePreblock(fields, mean, std)
for _ in range(3):
    x = pre(fields)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    x = pre(fields)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
gb = 2 * x.numel() * 4 / 1e9
ref = torch.cat([(fields["era5"][k] - 0.5) / 2.0 for k in pre.keys], dim=1)
print(f"wx_pre_apply C3 input ({x.shape[1]} ch x {H}x{W}): {ms * 1e3:.1f} us, {gb / (ms * 1e-3):.0f} GB/s (read + write {gb:.2f} GB); "
      f"max |x - torch| = {float((x - ref).abs().max()):.1e}")
t0 = time.perf_counter()
for _ in range(10):
    ref = torch.cat([(fields["era5"][k] - 0.5) / 2.0 for k in pre.keys], dim=1)
torch.cuda.synchronize()
print(f"torch eager (14 normalise kernels + cat): {(time.perf_counter() - t0) / 10 * 1e6:.1f} us")
