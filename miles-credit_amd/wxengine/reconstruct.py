"""Gen-2 split / re-flatten of the model output, mirrors of credit/postblock/reconstruct.py:

* `Reconstruct` (:25-83)       y_pred [B, C(, T), H, W] -> nested dict of VIEWS [B, n_levels, n_time, H, W] by
                                `metadata["target"]["_channel_map"]`; nothing is copied, exactly like the reference's slices.
* `FlattenToTensor` (:86-160)  the inverse: the named tensors, optionally forward-scaled (physical -> normalised,
                                (t - mean) / std per variable and level), concatenated in channel-map order.  Scaling and
                                concatenation are ONE pass of the engine's `wx_pre_*` kernel (the same fused
                                normalise + concatenate that serves the input side); no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional

from .preblock import DevicePreblock


class Reconstruct:
    def __init__(self, detach: bool = True, in_key: str = "y_pred", out_key: str = "y_processed"):
        self.detach, self.in_key, self.out_key = detach, in_key, out_key

    def __call__(self, batch_dict: dict) -> dict:
        y = batch_dict[self.in_key]
        cmap = batch_dict["metadata"]["target"]["_channel_map"]
        if y.dim() == 5:
            y = y.flatten(1, 2)
        out: Dict[str, Dict] = {}
        for key, info in cmap.items():
            t = y[:, info["slice"], ...]
            if self.detach:
                t = t.detach()
            out.setdefault(key.split("/")[0], {})[key] = t.unflatten(1, tuple(info["orig_shape"]))
        batch_dict[self.out_key] = out
        return batch_dict

    forward = __call__


class FlattenToTensor:
    """`mean` / `std`: {varname: scalar or per-level vector} in place of the reference's bridgescaler file (method
    "transform"); None = flatten as is."""

    def __init__(self, mean: Optional[Dict] = None, std: Optional[Dict] = None, key: str = "y_processed", out_key: str = "y_pred"):
        self.mean, self.std, self.key, self.out_key = mean, std, key, out_key
        self._pre = None
        self._order = None

    def __call__(self, batch_dict: dict) -> dict:
        nested = batch_dict[self.key]
        cmap = batch_dict["metadata"]["target"]["_channel_map"]
        order = sorted(cmap, key=lambda k: cmap[k]["slice"].start)
        flat = {k: nested[k.split("/")[0]][k] for k in order}
        if self._pre is None or self._order != order:
            # DevicePreblock orders by the reference's input sort key; the target map's own order is what counts here, so
            # the variables are presented under keys that already sort in channel-map order
            self._alias = {k: f"y/prognostic/3d/{i:04d}:{k.split('/')[-1]}" for i, k in enumerate(order)}
            mean = None if self.mean is None else {f"{i:04d}:{k.split('/')[-1]}": self.mean[k.split("/")[-1]]
                                                   for i, k in enumerate(order) if k.split("/")[-1] in self.mean}
            std = None if self.std is None else {f"{i:04d}:{k.split('/')[-1]}": self.std[k.split("/")[-1]]
                                                 for i, k in enumerate(order) if k.split("/")[-1] in self.std}
            self._pre = DevicePreblock({"y": {self._alias[k]: flat[k] for k in order}}, mean, std)
            self._order = order
        x = self._pre({"y": {self._alias[k]: flat[k] for k in order}})    # [B, C, T, H, W]
        batch_dict[self.out_key] = x.flatten(1, 2)                        # (B, C*T, H, W) like the reference
        return batch_dict

    forward = __call__
