"""The autoregressive hot loop of credit/applications/rollout_to_netcdf.py:274-310 on the engine.

    for step in 1..n: y = model(x) [+tracer fixer]; y_phys = y*std+mean; x = update_x(x, frc_t, y)

All tensors stay in HBM: forcing for every step is pre-staged on the device, y_phys is written into a
caller-provided ring so the host can drain it asynchronously (the reference does `.cpu().numpy()` every
step, rollout_to_netcdf.py:292), and the next input is assembled by the engine's tail kernel.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .engine import WXEngine


def channel_layout(cfg, n_static: int, n_dyn: int):
    """Single-source layout (credit/datasets/gen_2/channel_utils.py:161-250): x = [prog | static | dyn]."""
    n_prog = cfg.channels * cfg.levels + cfg.surface_channels
    if n_prog + n_static + n_dyn != cfg.base_input_channels:
        raise ValueError("n_prog + n_static + n_dyn must equal the model's input channels")
    return n_prog, n_static, n_dyn


def rollout(engine: WXEngine, x0: torch.Tensor, forcings: Sequence[Optional[torch.Tensor]],
            keep_phys: bool = True, keep_y: bool = False):
    """Run len(forcings) forecast steps. forcings[t] feeds the input of step t+2 (None on the last step).

    Returns (y_list, y_phys_list); lists are empty when not kept (bench mode keeps only the last)."""
    x = x0
    ys: List[torch.Tensor] = []
    phys: List[torch.Tensor] = []
    n = len(forcings)
    for t in range(n):
        frc = forcings[t]
        last = t == n - 1
        y, yp, xn = engine.step(x, frc, want_y=keep_y, want_phys=True, want_next=not last or frc is not None)
        if keep_y:
            ys.append(y)
        if keep_phys:
            phys.append(yp)
        elif last:
            phys.append(yp)
        if xn is not None:
            x = xn
    return ys, phys
