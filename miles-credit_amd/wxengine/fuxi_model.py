"""Registry surface for the FuXi engine: an `nn.Module` (a `credit.models.base_model.BaseModel` when the reference is importable)
with the constructor kwargs of `credit.models.fuxi.Fuxi` (fuxi.py:327-356) and a state dict under the reference's key names, so that

    from wxengine.fuxi_model import register
    register("fuxi_hip")                       # credit.models.register_model (credit/models/__init__.py:128-161)
    model = credit.models.load_model(conf)     # conf["model"]["type"] == "fuxi_hip"   (:301-387)

builds it like any other model and `BaseModel.load_model` (base_model.py:57-87) fills it from a checkpoint.  `forward(x)` runs
`wxengine.fuxi.FuxiHIP` (C ABI `wx_fuxi_*`) on the device of `x`; the weights are pushed to the engine on the first call after a
`load_state_dict`.  The stage keys are timm's (`blocks.<i>.attn.{logit_scale,q_bias,v_bias,cpb_mlp.0.*,cpb_mlp.2.*,qkv.*,proj.*}`, ...), i.e.
those of a reference FuXi checkpoint; `stage: "cr"` in the model section selects the V2-Cr block of credit/models/swin.py instead.
No CPU fallback.
"""
from __future__ import annotations

import copy
import logging
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .engine import WXEngineError
from .fuxi import FuxiConfig, FuxiHIP

logger = logging.getLogger(__name__)

try:  # drop-in: be a real BaseModel when the reference is installed
    from credit.models.base_model import BaseModel as _Base  # type: ignore
except Exception:  # noqa: BLE001 - any import problem means "reference not installed"
    _Base = nn.Module


def _mangle(key: str) -> str:
    return key.replace(".", "::")


class FuxiHIPModel(_Base):
    def __init__(self, precision: str = "bf16", **model_conf):
        super().__init__()
        self.cfg = cfg = FuxiConfig.from_model_conf(copy.deepcopy(model_conf))
        self.precision = precision
        # attributes the reference class exposes (fuxi.py:358-428)
        self.use_interp, self.use_spectral_norm = cfg.interp, cfg.use_spectral_norm
        self.use_padding = self.use_post_block = False
        self.img_size = self.img_size_original = (cfg.frames, cfg.image_height, cfg.image_width)
        self.patch_size = (cfg.frame_patch_size, cfg.patch_height, cfg.patch_width)
        self.input_resolution = (cfg.patches[0] // 2, cfg.patches[1] // 2)
        self.out_chans = cfg.out_chans
        self.channels, self.surface_channels, self.levels = cfg.channels, cfg.surface_channels, cfg.levels
        self._spec = cfg.state_spec()
        self._store = nn.ParameterDict()
        for key, shape in self._spec.items():
            self._store[_mangle(key)] = nn.Parameter(torch.zeros(shape, dtype=torch.float32), requires_grad=False)
        self._impl = None
        self._dirty = True

    # ---- state dict with reference key names -----------------------------------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False, **kwargs):
        out = OrderedDict() if destination is None else destination
        for key in self._spec:
            p = self._store[_mangle(key)]
            out[prefix + key] = p if keep_vars else p.detach()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """torch semantics: missing / unexpected key lists, size-mismatch RuntimeError; a uniform DDP `module.` prefix is dropped."""
        keys = list(state_dict.keys())
        if keys and all(k.startswith("module.") for k in keys):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        missing, errors = [], []
        for key in self._spec:
            if key not in state_dict:
                missing.append(key)
                continue
            src = state_dict[key]
            src = src.detach() if isinstance(src, torch.Tensor) else torch.as_tensor(np.asarray(src))
            dst = self._store[_mangle(key)]
            if tuple(src.shape) != tuple(dst.shape):
                errors.append(f"size mismatch for {key}: copying a param with shape {tuple(src.shape)} from checkpoint, "
                              f"the shape in current model is {tuple(dst.shape)}.")
                continue
            with torch.no_grad():
                dst.copy_(src.to(dst.dtype))
        from .swin import TIMM_DERIVED_SUFFIXES   # buffers older timm releases saved with the stage: derived data, not weights
        unexpected = [k for k in state_dict if k not in self._spec and not k.endswith(TIMM_DERIVED_SUFFIXES)]
        if strict and (missing or unexpected):
            errors.insert(0, f"Missing key(s) in state_dict: {missing[:8]}{' ...' if len(missing) > 8 else ''}; "
                             f"Unexpected key(s) in state_dict: {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''}.")
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(type(self).__name__, "\n\t".join(errors)))
        self._dirty = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- engine -------------------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, noise=None, forecast_step=None) -> torch.Tensor:
        if noise is not None:
            raise WXEngineError("FuxiHIPModel: noise injection is not implemented by the HIP engine")
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise WXEngineError("FuxiHIPModel has no CPU fallback: x must be a GPU tensor")
        dev = x.device.index or 0
        if self._impl is None or self._impl.device != dev:
            self._impl = FuxiHIP(precision=self.precision, device=dev, cfg=self.cfg)
            self._dirty = True
        if self._dirty:
            self._impl.load_state_dict({k: self._store[_mangle(k)].detach().cpu().numpy() for k in self._spec})
            self._dirty = False
        return self._impl(x.float())


def _standalone_load_model(cls, conf):
    """Checkpoint loader for installations WITHOUT the reference package (the contract of base_model.py:57-87, checkpoint.py:25-31)."""
    import os
    root = os.path.expandvars(conf["save_loc"])
    found = next((f for f in (os.path.join(root, n) for n in ("model_checkpoint.pt", "checkpoint.pt")) if os.path.isfile(f)), None)
    if found is None:
        raise ValueError(f"no model_checkpoint.pt / checkpoint.pt under {root}")
    blob = torch.load(found, map_location="cpu")
    model = cls(**{k: v for k, v in copy.deepcopy(conf["model"]).items() if k != "type"})
    result = model.load_state_dict(blob.get("model_state_dict", blob), strict=False)
    if result.unexpected_keys:
        raise RuntimeError(f"{found}: keys the model does not have: {list(result.unexpected_keys)[:8]}")
    if result.missing_keys:
        logger.warning("%s: %d model key(s) absent from the checkpoint (left at their initial values)", found, len(result.missing_keys))
    return model


if _Base is nn.Module:
    FuxiHIPModel.load_model = classmethod(_standalone_load_model)


def register(model_type: str = "fuxi_hip"):
    """Register the class with the reference's registry (credit.models.register_model)."""
    from credit.models import register_model  # type: ignore
    return register_model(model_type, "Loading the MI355X-native FuXi engine ...")(FuxiHIPModel)
