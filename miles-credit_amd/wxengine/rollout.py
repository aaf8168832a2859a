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


_INPUT_FIELD_TYPES = ("prognostic", "static", "dynamic_forcing")   # canonical concat rank (channel_utils.py:88-93)


def build_channel_layout(conf):
    """Mirror of credit/datasets/gen_2/channel_utils.py:161-250 for any number of data sources.

    Returns (groups, n_pred): groups = [(field_type, x_start, src_start or None, count), ...] in input-channel order, one per
    (source, field_type) run -- with several sources a field type's channels are contiguous only within a source -- ready for
    WXEngine.set_layout_groups; n_pred = number of prognostic channels.  Source blocks of y run prognostic-then-diagnostic, the
    forcing tensor holds the dynamic-forcing channels source-major.  Raises ValueError like the reference (history_len != 1,
    3-D variables without a level count)."""
    data = conf["data"]
    groups, x_cur, pred_cur, dyn_cur = [], 0, 0, 0
    for name, src in data["source"].items():
        src = src or {}
        variables = src.get("variables") or {}
        levels = src.get("levels")
        n_levels = len(levels) if levels else int((conf.get("model") or {}).get("levels") or 0)
        if src.get("history_len", data.get("history_len", 1)) != 1:
            raise ValueError(f"build_channel_layout: source '{name}' has history_len != 1; the flat rollout cannot shift a history window")

        def width(ft):
            grp = variables.get(ft) or {}
            n3, n2 = len(grp.get("vars_3D") or []), len(grp.get("vars_2D") or [])
            if n3 and not n_levels:
                raise ValueError(f"build_channel_layout: source '{name}' defines 3D variables but no level count is set")
            return n3 * n_levels + n2
        prog_src = pred_cur
        pred_cur += width("prognostic") + width("diagnostic")
        for ft in _INPUT_FIELD_TYPES:
            w = width(ft)
            if w == 0:
                continue
            if ft == "prognostic":
                s0 = prog_src
            elif ft == "dynamic_forcing":
                s0 = dyn_cur
                dyn_cur += w
            else:
                s0 = None
            groups.append((ft, x_cur, s0, w))
            x_cur += w
    return groups, sum(g[3] for g in groups if g[0] == "prognostic")


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
