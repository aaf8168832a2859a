"""Gen-2 named-tensor forecast loop on the device: the step loop of
credit/trainers/rollout_utils.py::run_forecast (:204-319) and the routing contract of ::assemble_rollout_batch (:322-430).

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


# where the next step's copy of an input variable comes from
FROM_PREDICTION, FROM_FORCING, FROM_IC = 0, 1, 2
_SOURCE_OF_FIELD_TYPE = {"prognostic": FROM_PREDICTION, "diagnostic": FROM_PREDICTION, "dynamic_forcing": FROM_FORCING}


class RolloutRouter:
    """The routing table of one forecast, built ONCE from the initial condition's key list.

    The contract is the one of credit/trainers/rollout_utils.py:322-430 (`assemble_rollout_batch`): the next input has exactly
    the IC's sources and variable keys, in the IC's order; `<source>/prognostic|diagnostic/...` keys take this step's
    processed prediction, `.../dynamic_forcing/...` keys take the incoming batch, everything else (static) keeps the IC
    tensor; a key its source does not provide falls back to the IC (newest time level when the IC carries a history) with a
    warning.  The reference re-derives this from the key strings at every step; here a step is one pass over a flat tuple of
    (source, key, origin) rows -- on the device loop that is all the host does between two kernels."""

    def __init__(self, ic_input: dict, history_len: int = 1):
        self.history_len = int(history_len)
        self.sources = [src for src, variables in ic_input.items() if variables]
        self.rows = tuple((src, key, self._origin(key)) for src in self.sources for key in ic_input[src])
        self._warned = set()

    @staticmethod
    def _origin(key: str) -> int:
        field_type = key.split("/", 2)[1] if key.count("/") else ""
        return _SOURCE_OF_FIELD_TYPE.get(field_type, FROM_IC)

    def _ic_value(self, tensor):
        if self.history_len > 1 and tensor.dim() >= 3 and tensor.shape[2] > 1:
            return tensor[:, :, -1:, ...]
        return tensor

    def _fallback(self, key, origin, ic_tensor):
        if key not in self._warned:   # once per key, not once per step
            self._warned.add(key)
            what = "y_processed" if origin == FROM_PREDICTION else "the forcing batch"
            logger.warning("rollout routing: '%s' is missing from %s; the initial-condition value is carried forward.", key, what)
        return self._ic_value(ic_tensor)

    def __call__(self, prediction: dict, forcing_input: dict, ic_input: dict) -> dict:
        out = {src: {} for src in self.sources}
        for src, key, origin in self.rows:
            if origin == FROM_IC:
                out[src][key] = self._ic_value(ic_input[src][key])
                continue
            pool = (prediction if origin == FROM_PREDICTION else forcing_input).get(src) or {}
            out[src][key] = pool[key] if key in pool else self._fallback(key, origin, ic_input[src][key])
        return out


def assemble_rollout_batch(full_data_dict: dict, curr_batch: dict, history_len: int = 1) -> dict:
    """Drop-in for the reference function of the same name (rollout_utils.py:322-430): `{"input": next input, "target": ...}`.
    Pure like the reference's: nothing is written into the caller's state dict (the routing table costs microseconds to build;
    `run_forecast` builds it once per forecast and does not come through here)."""
    prediction = full_data_dict["y_processed"]
    if not isinstance(prediction, dict):
        raise TypeError(f"y_processed is a {type(prediction).__name__}, not {{source: {{variable key: tensor}}}}: put Reconstruct "
                        "first in the post-block chain before rolling out more than one step")
    ic_input = full_data_dict["ic_preprocessed"]["input"]
    router = RolloutRouter(ic_input, history_len)
    return {"input": router(prediction, curr_batch.get("input") or {}, ic_input), "target": curr_batch.get("target")}


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
    pre = DevicePreblock(ic_batch["input"], mean, std)   # lives on the device of the IC tensors
    router = RolloutRouter(ic_batch["input"])
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
                nxt = router(full["y_processed"], next(it).get("input") or {}, ic_batch["input"])
                full["x_physical"] = nxt
                full["x"] = pre(nxt)
    return full
