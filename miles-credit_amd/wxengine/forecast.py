"""Gen-2 named-tensor forecast loop on the device: mirrors of
credit/trainers/rollout_utils.py::assemble_rollout_batch (:322-430) and ::run_forecast (:204-319).

Per step, with every tensor resident in HBM:
    x --model--> y_pred --Reconstruct--> named y (views) --InverseScale--> physical --fixers--> y_processed --consume-->
    assemble_rollout_batch(prognostic/diagnostic <- y_processed, dynamic_forcing <- this step's batch, static <- IC)
    --DevicePreblock (normalise + concatenate, one kernel)--> x
The chain of post blocks is a list of callables on the batch dict, exactly like the reference's `apply_postblocks`.
"""
from __future__ import annotations

import logging
from typing import Callable, Dict, Iterable, List, Optional

from .preblock import DevicePreblock
from .reconstruct import Reconstruct

logger = logging.getLogger(__name__)


def assemble_rollout_batch(full_data_dict: dict, curr_batch: dict, history_len: int = 1) -> dict:
    """rollout_utils.py:322-430: route every IC variable key to its source for the next step's input."""
    pred = full_data_dict["y_processed"]
    ic = full_data_dict["ic_preprocessed"]
    if not isinstance(pred, dict):
        raise TypeError("assemble_rollout_batch: full_data_dict['y_processed'] must be a nested dict {source: {var_key: tensor}}. "
                        "For multi-step rollout, 'Reconstruct' must be the first postblock. "
                        f"Got {type(pred).__name__}.")

    def newest(t):
        if history_len > 1 and t.dim() >= 3 and t.shape[2] > 1:
            return t[:, :, -1:, ...]
        return t

    out: Dict[str, Dict] = {}
    for source, variables in ic["input"].items():
        if not variables:
            continue
        out[source] = {}
        cur = curr_batch.get("input", {}).get(source, {})
        prd = pred.get(source, {})
        for key, ic_t in variables.items():
            parts = key.split("/")
            ft = parts[1] if len(parts) > 1 else ""
            if ft in ("prognostic", "diagnostic"):
                if key in prd:
                    out[source][key] = prd[key]
                else:
                    logger.warning("assemble_rollout_batch: '%s' not in y_processed; carrying forward from ic_preprocessed.", key)
                    out[source][key] = newest(ic_t)
            elif ft == "dynamic_forcing":
                if key in cur:
                    out[source][key] = cur[key]
                else:
                    logger.warning("assemble_rollout_batch: dynamic_forcing '%s' not in curr_batch; carrying forward from "
                                   "ic_preprocessed.", key)
                    out[source][key] = newest(ic_t)
            else:
                out[source][key] = newest(ic_t)
    return {"input": out, "target": curr_batch.get("target")}


class InverseScale:
    """The inverse scaler post block: physical = normalised * std + mean per variable (scalar or per-level statistics);
    variables without statistics pass through.  Operates on `batch_dict["y_processed"]` in place of the bridgescaler call."""

    def __init__(self, mean: Dict, std: Dict):
        self.mean, self.std = mean, std

    def __call__(self, batch_dict: dict) -> dict:
        import torch
        for source, variables in batch_dict["y_processed"].items():
            for key in list(variables.keys()):
                name = key.split("/")[-1]
                if name not in self.mean:
                    continue
                t = variables[key]
                m = torch.as_tensor(self.mean[name], dtype=t.dtype, device=t.device).reshape(1, -1, 1, 1, 1)
                s = torch.as_tensor(self.std[name], dtype=t.dtype, device=t.device).reshape(1, -1, 1, 1, 1)
                variables[key] = t * s + m
        return batch_dict


def run_forecast(model, ic_batch: dict, forcing_batches: Iterable[dict], n_steps: int, target_channel_map: Dict,
                 mean: Optional[Dict], std: Optional[Dict], step_postblocks: List[Callable[[dict], dict]],
                 consume: Callable[[dict, int], None]) -> dict:
    """rollout_utils.py:204-319 with the engine's device blocks.  `ic_batch["input"]` / the forcing batches hold PHYSICAL named
    tensors [B, n_levels, T, H, W]; `step_postblocks` runs after `Reconstruct` (e.g. InverseScale, the conservation fixers);
    `consume(y_processed, step)` stands in for `save_output_fn`.  Returns the final state dict."""
    import torch
    full: dict = {"ic_preprocessed": {"input": ic_batch["input"]}, "x_physical": ic_batch["input"],
                  "metadata": {"target": {"_channel_map": target_channel_map}}}
    pre = DevicePreblock(ic_batch["input"], mean, std)
    full["metadata"]["input"] = {"_channel_map": pre.channel_map}
    full["x"] = pre(ic_batch["input"])
    rec = Reconstruct()
    it = iter(forcing_batches)
    with torch.no_grad():
        for step in range(1, n_steps + 1):
            full["y_pred"] = model(full["x"])
            full = rec(full)
            for blk in step_postblocks:
                full = blk(full)
            consume(full["y_processed"], step)
            if step < n_steps:
                nxt = assemble_rollout_batch(full, next(it))
                full["x_physical"] = nxt["input"]
                full["x"] = pre(nxt["input"])
    return full
