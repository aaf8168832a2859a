"""Host-side mirror of the reference model interface for the hot path.

`WXFormerHIP` takes the same constructor kwargs as the reference
`credit.models.crossformer.CrossFormer` (credit/models/crossformer.py:372-401), exposes a
state dict with the reference's key names (so reference checkpoints load unchanged,
credit/models/base_model.py:57-87), and implements `forward(x)` by calling the HIP engine
through the C ABI.  When the reference package is importable it subclasses
`credit.models.base_model.BaseModel`, so it can be registered with
`credit.models.register_model("crossformer_hip")` and selected by `model.type` in the YAML
(credit/models/__init__.py:128-161) — see INTEGRATION.md.

There is no CPU fallback: constructing the engine without the HIP library or a GPU raises.
"""
from __future__ import annotations

import copy
import logging
import os
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from .config import WXConfig
from .engine import WXEngine, WXEngineError

logger = logging.getLogger(__name__)

try:  # drop-in: be a real BaseModel when the reference is installed
    from credit.models.base_model import BaseModel as _Base  # type: ignore
except Exception:  # noqa: BLE001 - any import problem means "reference not installed"
    _Base = nn.Module


def _mangle(key: str) -> str:
    return key.replace(".", "::")


class WXFormerHIP(_Base):
    """CrossFormer forward on MI355X. Same kwargs as the reference class; extra kwarg `precision`."""

    def __init__(self, precision: str = "bf16", arch: str = "crossformer", **model_conf):
        super().__init__()
        model_conf = copy.deepcopy(model_conf)
        self.cfg = WXConfig.from_model_conf(model_conf, arch=arch)
        self.precision = precision
        cfg = self.cfg
        # attributes the reference's callers read (SURVEY.md §8(b) "Constructor")
        self.image_height, self.image_width = cfg.image_height, cfg.image_width
        self.frames, self.output_frames = cfg.frames, cfg.output_frames
        self.channels, self.levels, self.surface_channels = cfg.channels, cfg.levels, cfg.surface_channels
        self.use_padding, self.use_interp = cfg.pad_activate, cfg.interp
        self.use_spectral_norm = cfg.use_spectral_norm
        self.use_post_block = bool(cfg.post_conf.get("activate", False))
        self._spec = cfg.state_spec()
        self._store = nn.ParameterDict()
        for key, shape in self._spec.items():
            self._store[_mangle(key)] = nn.Parameter(torch.zeros(shape, dtype=torch.float32), requires_grad=False)
        self._engine: Optional[WXEngine] = None
        self._dirty = True
        self._denorm = None
        self._tracer = self._tracer_from_post_conf(cfg.post_conf)

    # ---- post block ---------------------------------------------------------------------------------------
    @staticmethod
    def _tracer_from_post_conf(post_conf: Dict):
        """In-model TracerFixer (fused into the engine's tail kernel).  SKEBS and hybrid-sigma grids are rejected."""
        if not post_conf or not post_conf.get("activate", False):
            return None
        if (post_conf.get("skebs") or {}).get("activate", False):
            raise ValueError("post_conf.skebs is not implemented by the HIP engine")
        tf = post_conf.get("tracer_fixer") or {}
        if not tf.get("activate", False):
            return None
        if "tracer_inds" not in tf:
            raise ValueError("post_conf.tracer_fixer.tracer_inds missing (the reference's parser injects it)")
        return dict(inds=list(tf["tracer_inds"]), thres=list(tf["tracer_thres"]),
                    thres_max=tf.get("tracer_thres_max"), denorm=bool(tf.get("denorm", False)))

    def _global_fixers(self):
        """[(name, conf)] of in-model global fixers, in PostBlock order (credit/postblock/gen1.py:56-99)."""
        pc = self.cfg.post_conf or {}
        out = []
        if not pc.get("activate", False):
            return out
        for name in ("global_mass_fixer", "global_water_fixer", "global_energy_fixer", "global_energy_fixer_updown"):
            sub = pc.get(name) or {}
            if sub.get("activate", False) and not sub.get("activate_outside_model", False):
                if sub.get("grid_type", "pressure") not in ("pressure", "sigma"):
                    raise ValueError(f"post_conf.{name}: grid_type must be 'pressure' or 'sigma'")
                out.append((name, sub))
        return out

    def set_physics(self, lat2d, lon2d, p_levels=None, gph_surf=None, mean_in=None, std_in=None, coef_a=None, coef_b=None):
        """What the reference reads from `post_conf.data.save_loc_physics` (lat/lon, pressure levels OR the hybrid
        coefficients a/b of `grid_type: sigma`, surface geopotential) and, for `denorm: True` fixers, the INPUT-channel
        statistics (the output ones come from set_denorm)."""
        self._physics = dict(lat2d=np.asarray(lat2d, np.float32), lon2d=np.asarray(lon2d, np.float32),
                             p=None if p_levels is None else np.asarray(p_levels, np.float32),
                             coef_a=None if coef_a is None else np.asarray(coef_a, np.float32),
                             coef_b=None if coef_b is None else np.asarray(coef_b, np.float32),
                             gph=None if gph_surf is None else np.asarray(gph_surf, np.float32),
                             mean_in=mean_in, std_in=std_in)
        self._dirty = True

    def _build_post(self, device_index: int):
        from .engine import WXPostBlock
        fixers = self._global_fixers()
        if not fixers:
            return None
        ph = getattr(self, "_physics", None)
        if ph is None:
            raise WXEngineError("global fixers are active: call set_physics(lat2d, lon2d, p_levels, gph_surf) first")
        cfg = self.cfg
        pb = WXPostBlock(cfg.out_hw[0], cfg.out_hw[1], cfg.base_input_channels, cfg.frames, cfg.base_output_channels,
                         device_index)
        midpoint = bool(fixers[0][1].get("midpoint", False))
        sigma = fixers[0][1].get("grid_type", "pressure") == "sigma"
        if any((c.get("grid_type", "pressure") == "sigma") != sigma for _, c in fixers):
            raise ValueError("all global fixers must agree on `grid_type`")
        if sigma:
            if ph["coef_a"] is None or ph["coef_b"] is None:
                raise WXEngineError("grid_type sigma: set_physics(..., coef_a=, coef_b=) first")
            sp = {int(c["sp_inds"]) for _, c in fixers}
            if len(sp) != 1:
                raise ValueError("all global fixers must name the same surface-pressure channel (`sp_inds`)")
            pb.set_grid_sigma(ph["lat2d"], ph["lon2d"], ph["coef_a"], ph["coef_b"], sp.pop(), midpoint)
        else:
            if ph["p"] is None:
                raise WXEngineError("grid_type pressure: set_physics(..., p_levels=) first")
            pb.set_grid(ph["lat2d"], ph["lon2d"], ph["p"], midpoint)
        if any(c.get("denorm", False) for _, c in fixers):
            if self._denorm is None or ph["mean_in"] is None:
                raise WXEngineError("denorm fixers need set_denorm(mean, std) and set_physics(..., mean_in=, std_in=)")
            pb.set_stats(ph["mean_in"], ph["std_in"], self._denorm[0], self._denorm[1])
        n_seconds = 3600.0 * float(((cfg.post_conf.get("data") or {}).get("lead_time_periods", 6)))
        for name, c in fixers:
            dn = bool(c.get("denorm", False))
            if bool(c.get("midpoint", False)) != midpoint:
                raise ValueError("all global fixers must agree on `midpoint`")
            if name == "global_mass_fixer":
                pb.add_mass_fixer(int(c["q_inds"][0]), int(c["fix_level_num"]), dn)
            elif name == "global_water_fixer":
                pb.add_water_fixer(int(c["q_inds"][0]), int(c["precip_ind"]), int(c["evapor_ind"]), n_seconds, dn)
            elif name == "global_energy_fixer_updown":
                if ph["gph"] is None:
                    raise WXEngineError("global_energy_fixer_updown needs the surface geopotential: set_physics(..., gph_surf=)")
                flux = [int(c[k]) for k in ("TOA_down_solar_ind", "TOA_up_solar_ind", "TOA_up_OLR_ind", "surf_down_solar_ind",
                                            "surf_up_solar_ind", "surf_down_LW_ind", "surf_up_LW_ind", "surf_SH_ind", "surf_LH_ind")]
                pb.add_energy_fixer_updown(int(c["T_inds"][0]), int(c["q_inds"][0]), int(c["U_inds"][0]), int(c["V_inds"][0]),
                                           flux, ph["gph"], n_seconds, dn)
            else:
                if ph["gph"] is None:
                    raise WXEngineError("global_energy_fixer needs the surface geopotential: set_physics(..., gph_surf=)")
                rad = [int(c["TOA_rad_inds"][0]), int(c["TOA_rad_inds"][1]), int(c["surf_rad_inds"][0]),
                       int(c["surf_rad_inds"][1]), int(c["surf_flux_inds"][0]), int(c["surf_flux_inds"][1])]
                pb.add_energy_fixer(int(c["T_inds"][0]), int(c["q_inds"][0]), int(c["U_inds"][0]), int(c["V_inds"][0]),
                                    rad, ph["gph"], n_seconds, dn)
        return pb

    def set_denorm(self, mean, std):
        """Per-output-channel statistics (what the reference reads from its scaler files)."""
        self._denorm = (np.asarray(mean, dtype=np.float32).ravel(), np.asarray(std, dtype=np.float32).ravel())
        self._dirty = True

    # ---- state dict with reference key names -------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False, **kwargs):
        out = OrderedDict() if destination is None else destination
        for key in self._spec:
            p = self._store[_mangle(key)]
            out[prefix + key] = p if keep_vars else p.detach()
        return out

    def _legacy_aliases(self):
        """{legacy key: spec key} for `wxformer` CrossEmbed parameters.  Checkpoints written before the convs were wrapped
        in ZeroPad2d keep them at `...convs.<i>.<suffix>`; the reference moves them to `...convs.<i>.1.<suffix>` while
        loading (credit/models/wxformer/crossformer.py:247-283, 335-352)."""
        out = {}
        if self.cfg.arch != "wxformer":
            return out
        for key in self._spec:
            parts = key.split(".")
            for i in range(len(parts) - 3):
                if parts[i] == "convs" and parts[i + 1].isdigit() and parts[i + 2] == "1":
                    out[".".join(parts[:i + 2] + parts[i + 3:])] = key
        return out

    @staticmethod
    def _same_layout(a, b) -> bool:
        """Shapes agree up to singleton dimensions ((1, C, 1, 1) vs (C,)); anything else is torch's "size mismatch"."""
        return [d for d in a if d != 1] == [d for d in b if d != 1]

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """torch semantics (missing / unexpected lists, size-mismatch RuntimeError) plus the two rewrites reference
        checkpoints need: a uniform DistributedDataParallel `module.` prefix is dropped and legacy CrossEmbed keys migrate."""
        keys = list(state_dict.keys())
        if keys and all(k.startswith("module.") for k in keys):
            state_dict = {k[len("module."):]: v for k, v in state_dict.items()}
        alias = self._legacy_aliases()
        source, migrated = {}, 0
        for k, v in state_dict.items():
            tgt = alias.get(k)
            if tgt is not None and tgt not in state_dict:   # never clobber a key the checkpoint already has in the new layout
                source[tgt] = v
                migrated += 1
            else:
                source[k] = v
        if migrated:
            logger.warning("Legacy checkpoint: remapped %d CrossEmbedLayer conv key(s) (convs.<i>.X -> convs.<i>.1.X).", migrated)
        missing, errors = [], []
        for key in self._spec:
            if key not in source:
                missing.append(key)
                continue
            src = source[key]
            src = src.detach() if isinstance(src, torch.Tensor) else torch.as_tensor(np.asarray(src))
            dst = self._store[_mangle(key)]
            if not self._same_layout(tuple(src.shape), tuple(dst.shape)):
                errors.append(f"size mismatch for {key}: copying a param with shape {tuple(src.shape)} from checkpoint, "
                              f"the shape in current model is {tuple(dst.shape)}.")
                continue
            with torch.no_grad():
                dst.copy_(src.reshape(dst.shape).to(dst.dtype))
        unexpected = [k for k in source if k not in self._spec]
        if strict and (missing or unexpected):
            errors.insert(0, f"Missing key(s) in state_dict: {missing[:8]}{' ...' if len(missing) > 8 else ''}; "
                             f"Unexpected key(s) in state_dict: {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''}.")
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(type(self).__name__, "\n\t".join(errors)))
        self._dirty = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- engine ---------------------------------------------------------------------------------------
    def _sync_engine(self, device: torch.device):
        if self._engine is None:
            self._engine = WXEngine(self.cfg, self.precision, device.index or 0)
            self._dirty = True
        if self._dirty:
            self._engine.load_state_dict({k: self._store[_mangle(k)].detach().cpu().numpy() for k in self._spec})
            self._engine.finalize()
            if self._denorm is not None:
                self._engine.set_denorm(*self._denorm)
            if self._tracer is not None:
                if self._tracer["denorm"] and self._denorm is None:
                    raise WXEngineError("tracer_fixer.denorm is True: call set_denorm(mean, std) first")
                self._engine.set_tracer_fixer(self._tracer["inds"], self._tracer["thres"], self._tracer["thres_max"],
                                              self._tracer["denorm"])
            self._engine.attach_postblock(self._build_post(device.index or 0))
            self._dirty = False
        return self._engine

    @property
    def engine(self) -> WXEngine:
        if self._engine is None:
            raise WXEngineError("engine not built yet: call the model once on a GPU tensor")
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise WXEngineError("WXFormerHIP runs only on the GPU (no CPU fallback); move the input to cuda")
        eng = self._sync_engine(x.device)
        return eng.forward(x.contiguous().float())


def _standalone_load_model(cls, conf):
    """Checkpoint loader for installations WITHOUT the reference package.  When `credit` is importable the class inherits
    `BaseModel.load_model` (credit/models/base_model.py:57-87) and this function is not attached; it keeps that method's
    contract: `<save_loc>/model_checkpoint.pt`, else `<save_loc>/checkpoint.pt`; `model_state_dict` or a bare state dict;
    non-strict load where unexpected keys are an error and missing keys a warning (credit/models/checkpoint.py:25-31)."""
    root = os.path.expandvars(conf["save_loc"])
    found = next((f for f in (os.path.join(root, n) for n in ("model_checkpoint.pt", "checkpoint.pt")) if os.path.isfile(f)), None)
    if found is None:
        raise ValueError(f"no model_checkpoint.pt / checkpoint.pt under {root}")
    blob = torch.load(found, map_location="cpu")
    kwargs = {k: v for k, v in copy.deepcopy(conf["model"]).items() if k != "type"}
    model = cls(**kwargs)
    result = model.load_state_dict(blob.get("model_state_dict", blob), strict=False)
    if result.unexpected_keys:
        raise RuntimeError(f"{found}: keys the model does not have: {list(result.unexpected_keys)[:8]}")
    if result.missing_keys:
        logger.warning("%s: %d model key(s) absent from the checkpoint (left at their initial values)", found, len(result.missing_keys))
    return model


if _Base is nn.Module:   # reference not installed: provide the entry point ourselves; otherwise BaseModel.load_model is inherited
    WXFormerHIP.load_model = classmethod(_standalone_load_model)


class WXFormerPSHIP(WXFormerHIP):
    """`model.type: wxformer` / `wxformer_base` (credit/models/wxformer/crossformer.py): PixelShuffle decoder."""

    def __init__(self, precision: str = "bf16", **model_conf):
        model_conf.pop("arch", None)
        super().__init__(precision=precision, arch="wxformer", **model_conf)


def register(model_type: str = "crossformer_hip", wxformer_type: str = "wxformer_hip"):
    """Register both engine classes with the reference's registry (credit.models.register_model)."""
    from credit.models import register_model  # type: ignore
    register_model(wxformer_type, "Loading the MI355X-native WXFormer (PixelShuffle decoder) engine ...")(WXFormerPSHIP)
    return register_model(model_type, "Loading the MI355X-native CrossFormer engine ...")(WXFormerHIP)
