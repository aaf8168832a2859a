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


_INPUT_ORDER = ("prognostic", "static", "dynamic_forcing")   # concat rank of the input tensor (channel_utils.py:88-93)
_OUTPUT_ORDER = ("prognostic", "diagnostic")                  # one source's block of the prediction tensor


def _source_widths(conf):
    """[(source name, {field type: channel count})] from `conf["data"]["source"]`: 3-D variables count once per level (the
    source's own level list, else `model.levels`), 2-D variables once."""
    data = conf["data"]
    default_levels = int((conf.get("model") or {}).get("levels") or 0)
    table = []
    for name, src in data["source"].items():
        src = src or {}
        if src.get("history_len", data.get("history_len", 1)) != 1:
            raise ValueError(f"build_channel_layout: source '{name}' has history_len != 1; the flat rollout cannot shift a history window")
        n_levels = len(src["levels"]) if src.get("levels") else default_levels
        widths = {}
        for field_type, grp in (src.get("variables") or {}).items():
            grp = grp or {}
            n3, n2 = len(grp.get("vars_3D") or []), len(grp.get("vars_2D") or [])
            if n3 and not n_levels:
                raise ValueError(f"build_channel_layout: source '{name}' defines 3D variables but no level count is set")
            widths[field_type] = n3 * n_levels + n2
        table.append((name, widths))
    return table


def build_channel_layout(conf):
    """The channel bookkeeping of credit/datasets/gen_2/channel_utils.py:161-250 for any number of data sources.

    Returns (groups, n_pred): groups = [(field_type, x_start, src_start or None, count), ...] in input-channel order, one per
    (source, field_type) run -- with several sources a field type's channels are contiguous only within a source -- ready for
    WXEngine.set_layout_groups; n_pred = number of prognostic channels.  Three tensors are described at once: the input x
    (per source: prognostic | static | dynamic_forcing), the prediction y (per source: prognostic | diagnostic) and the forcing
    tensor (every source's dynamic-forcing channels, source-major); a group's src_start is its offset in y (prognostic) or in
    the forcing tensor (dynamic_forcing).  Raises ValueError like the reference (history_len != 1, 3-D variables without a
    level count)."""
    table = _source_widths(conf)
    # exclusive prefix sums over the sources: where each source's block starts in y and in the forcing tensor
    y_start, frc_start, y_cur, frc_cur = {}, {}, 0, 0
    for name, w in table:
        y_start[name], frc_start[name] = y_cur, frc_cur
        y_cur += sum(w.get(ft, 0) for ft in _OUTPUT_ORDER)
        frc_cur += w.get("dynamic_forcing", 0)
    source_offset = {"prognostic": y_start, "dynamic_forcing": frc_start}
    groups, x_cur = [], 0
    for name, w in table:
        for ft in _INPUT_ORDER:
            n = w.get(ft, 0)
            if n:
                groups.append((ft, x_cur, source_offset[ft][name] if ft in source_offset else None, n))
                x_cur += n
    return groups, sum(n for ft, _, _, n in groups if ft == "prognostic")


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
