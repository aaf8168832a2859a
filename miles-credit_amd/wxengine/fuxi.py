"""BASELINE config 5: the FuXi forward on the engine (C ABI `wx_fuxi_*`, kernels in csrc/wx_fuxi.h).

`FuxiHIP` takes the constructor kwargs of credit/models/fuxi.py::Fuxi (:327-356) and a state dict with the reference's key names:
`cube_embedding.proj.*` (Conv3d, never spectral-normed), `cube_embedding.norm.*`, `u_transformer.down.{conv,b.0,b.1,b.3,b.4}.*`,
`u_transformer.up.*`, `fc.*` -- with `weight_orig / weight_u / weight_v` triples wherever `apply_spectral_norm` (:16-22) wrapped a
Conv2d / Linear / ConvTranspose2d (eval mode: W = weight_orig / (u . W_mat v), folded once on the host) -- and, under
`u_transformer.layer.blocks.{i}.`, the keys of a Swin V2-Cr block (credit/models/swin.py:330-502; see wxengine.swin.SwinStage).

The stage is the one deliberate difference: the reference builds timm's `SwinTransformerV2Stage` there (fuxi.py:250-260) and timm
is not vendored, so that class cannot be pinned; the engine runs the V2-Cr stage it can pin to the reference's own swin.py.
Everything around the stage follows fuxi.py's modules and is pinned to them (tests/test_fuxi.py).

Scope: padding_conf / post_conf off, no noise injection, image a multiple of the patch (the trailing bilinear resize to the input
size, :491-493, is then the identity), frame_patch_size == frames.  Anything else raises.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from .engine import PREC, WXEngineError, _check, load_library
from .swin import TIMM_DERIVED_SUFFIXES, effective_logit_scale, relative_position_bias, timm_block_tensors
from .synth import keyed_normal, power_iterate

META_HIDDEN = 384   # swin.py:233-239: the meta network's hidden width


class wx_fuxi_desc(C.Structure):
    _fields_ = [("precision", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C_in", C.c_int32), ("C_out", C.c_int32),
                ("frames", C.c_int32), ("patch_h", C.c_int32), ("patch_w", C.c_int32), ("dim", C.c_int32), ("heads", C.c_int32),
                ("window", C.c_int32), ("depth", C.c_int32), ("groups_down", C.c_int32), ("groups_up", C.c_int32), ("stage_variant", C.c_int32)]


STAGE_VARIANT = {"cr": 0, "timm": 1}   # WX_STAGE_V2_CR / WX_STAGE_TIMM_V2 (include/wxengine.h)
CPB_HIDDEN = 512                       # timm: cpb_mlp = Linear(2, 512) -> ReLU -> Linear(512, heads, bias=False)


def window_padding(n: int, window: int) -> Tuple[int, int]:
    """fuxi.py:31-65 get_pad3d for one axis: (front, back), the smaller half in front."""
    rem = n % window
    if not rem:
        return 0, 0
    tot = window - rem
    return tot // 2, tot - tot // 2


@dataclass
class FuxiConfig:
    image_height: int = 640
    patch_height: int = 16
    image_width: int = 1280
    patch_width: int = 16
    levels: int = 15
    frames: int = 2
    frame_patch_size: int = 2
    dim: int = 1536
    num_groups: object = 32
    channels: int = 4
    surface_channels: int = 7
    input_only_channels: int = 0
    output_only_channels: int = 0
    num_heads: int = 8
    depth: int = 48
    window_size: int = 7
    use_spectral_norm: bool = True
    interp: bool = True
    meta_hidden: int = META_HIDDEN
    # which block the stage is made of: "timm" = timm.models.swin_transformer_v2.SwinTransformerV2Stage, what the reference builds
    # (fuxi.py:4-5, 250-260) and what its checkpoints contain; "cr" = credit/models/swin.py's V2-Cr block (the variant whose goldens
    # come from reference code -- timm is not installable here, so the timm variant follows timm's published block: parity unpinned)
    stage: str = "timm"
    # timm stage only: does attn.qkv read `weight_orig` UN-normalised?  True reproduces what the reference module computes when it is built
    # on the CPU, loaded with load_state_dict and called without being moved or cast in between (see TIMM_UNNORMALISED below: timm's
    # WindowAttention reads `self.qkv.weight` without calling the module, so spectral_norm's pre-hook never runs and `weight` is still the
    # alias of `weight_orig`).  After `.to(device)` / `.half()` / any `module._apply`, torch rebuilds `weight` as a separate tensor holding
    # the LAST value the hook computed -- a maintainer who runs the reference that way (or who trains: the hook fires in train mode's
    # power iteration through other paths) sets this to False and gets `weight_orig / sigma` like every other layer.
    timm_qkv_unnormalised: bool = True

    @classmethod
    def from_model_conf(cls, conf: Dict) -> "FuxiConfig":
        conf = dict(conf)
        for k in ("padding_conf", "post_conf"):
            v = conf.pop(k, None)
            if v and v.get("activate", False):
                raise ValueError(f"FuxiHIP: {k} is not implemented by the HIP engine")
        if conf.pop("use_noise", False):
            raise ValueError("FuxiHIP: noise injection is not implemented by the HIP engine")
        for k in ("proj_drop", "attn_drop", "drop_path"):
            if conf.pop(k, 0):
                raise ValueError(f"FuxiHIP: {k} > 0 is a training-time option; the engine runs the eval forward")
        for k in ("noise_latent_dim", "noise_factor", "noise_scheduler", "freeze", "type"):
            conf.pop(k, None)
        known = {f for f in cls.__dataclass_fields__}
        extra = set(conf) - known
        if extra:   # the reference class swallows them in **kwargs (fuxi_6h_single_step.yml carries `pad_lon` / `pad_lat` there)
            import logging
            logging.getLogger(__name__).warning("FuxiHIP: ignoring model keys the reference ignores too: %s", sorted(extra))
        cfg = cls(**{k: v for k, v in conf.items() if k in known})
        cfg.check()
        return cfg

    def check(self) -> None:
        if self.stage not in STAGE_VARIANT:
            raise ValueError("FuxiHIP: stage must be 'timm' or 'cr'")
        if self.frame_patch_size != self.frames:
            raise ValueError("FuxiHIP: frame_patch_size must equal frames (fuxi.py:470 squeezes the time axis)")
        if self.image_height % self.patch_height or self.image_width % self.patch_width:
            raise ValueError("FuxiHIP: the image must be a multiple of the patch")
        if (self.image_height // self.patch_height) % 2 or (self.image_width // self.patch_width) % 2:
            raise ValueError("FuxiHIP: the patch grid must be even (DownBlock / UpBlock)")

    # fuxi.py:381-383
    @property
    def in_chans(self) -> int:
        return self.channels * self.levels + self.surface_channels + self.input_only_channels

    @property
    def out_chans(self) -> int:
        return self.channels * self.levels + self.surface_channels + self.output_only_channels

    @property
    def groups(self) -> Tuple[int, int]:
        g = self.num_groups
        return (int(g), int(g)) if np.isscalar(g) else (int(g[0]), int(g[1]))

    @property
    def patches(self) -> Tuple[int, int]:
        return self.image_height // self.patch_height, self.image_width // self.patch_width

    @property
    def stage_feat(self) -> Tuple[int, int]:
        """token map the stage sees: half the patch grid, zero-padded to a multiple of the window (fuxi.py:231-243)."""
        hd, wd = self.patches[0] // 2, self.patches[1] // 2
        return hd + sum(window_padding(hd, self.window_size)), wd + sum(window_padding(wd, self.window_size))

    def state_spec(self) -> "OrderedDict[str, tuple]":
        """Reference-layout state dict: key -> shape."""
        d, sn = self.dim, self.use_spectral_norm
        spec: "OrderedDict[str, tuple]" = OrderedDict()

        def wrapped(prefix, shape, transposed=False):   # a module apply_spectral_norm wraps (Conv2d / Linear / ConvTranspose2d)
            if sn:
                spec[prefix + ".bias"] = (shape[1] if transposed else shape[0],)
                spec[prefix + ".weight_orig"] = tuple(shape)
                rows = shape[1] if transposed else shape[0]
                spec[prefix + ".weight_u"] = (rows,)
                spec[prefix + ".weight_v"] = (int(np.prod(shape)) // rows,)
            else:
                spec[prefix + ".weight"] = tuple(shape)
                spec[prefix + ".bias"] = (shape[1] if transposed else shape[0],)

        def plain(prefix, n):
            spec[prefix + ".weight"] = (n,)
            spec[prefix + ".bias"] = (n,)

        spec["cube_embedding.proj.weight"] = (d, self.in_chans, self.frame_patch_size, self.patch_height, self.patch_width)
        spec["cube_embedding.proj.bias"] = (d,)
        plain("cube_embedding.norm", d)
        wrapped("u_transformer.down.conv", (d, d, 3, 3))
        for i in (0, 3):
            wrapped(f"u_transformer.down.b.{i}", (d, d, 3, 3))
            plain(f"u_transformer.down.b.{i + 1}", d)
        def wrapped_nobias(prefix, shape):   # nn.Linear(..., bias=False) under apply_spectral_norm
            if sn:
                spec[prefix + ".weight_orig"] = tuple(shape)
                spec[prefix + ".weight_u"] = (shape[0],)
                spec[prefix + ".weight_v"] = (int(np.prod(shape)) // shape[0],)
            else:
                spec[prefix + ".weight"] = tuple(shape)

        for i in range(self.depth):
            p = f"u_transformer.layer.blocks.{i}."
            if self.stage == "timm":
                # timm.models.swin_transformer_v2.SwinTransformerV2Block: attn = WindowAttention(logit_scale [heads, 1, 1], cpb_mlp,
                # qkv = Linear(dim, 3 dim, bias=False), q_bias, v_bias (k_bias / coordinate tables / attn_mask: non-persistent buffers),
                # proj), norm1, mlp = Mlp(fc1, fc2), norm2; apply_spectral_norm (fuxi.py:16-22) wraps every nn.Linear among them
                spec[p + "attn.logit_scale"] = (self.num_heads, 1, 1)
                spec[p + "attn.q_bias"] = (d,)
                spec[p + "attn.v_bias"] = (d,)
                wrapped(p + "attn.cpb_mlp.0", (CPB_HIDDEN, 2))
                wrapped_nobias(p + "attn.cpb_mlp.2", (self.num_heads, CPB_HIDDEN))
                wrapped_nobias(p + "attn.qkv", (3 * d, d))
                wrapped(p + "attn.proj", (d, d))
                plain(p + "norm1", d)
                wrapped(p + "mlp.fc1", (4 * d, d))
                wrapped(p + "mlp.fc2", (d, 4 * d))
                plain(p + "norm2", d)
                continue
            plain(p + "norm1", d)
            spec[p + "attn.logit_scale"] = (self.num_heads,)
            wrapped(p + "attn.qkv", (3 * d, d))
            wrapped(p + "attn.proj", (d, d))
            wrapped(p + "attn.meta_mlp.fc1", (self.meta_hidden, 2))
            wrapped(p + "attn.meta_mlp.fc2", (self.num_heads, self.meta_hidden))
            plain(p + "norm2", d)
            wrapped(p + "mlp.fc1", (4 * d, d))
            wrapped(p + "mlp.fc2", (d, 4 * d))
        wrapped("u_transformer.up.conv", (2 * d, d, 2, 2), transposed=True)
        for i in (0, 3):
            wrapped(f"u_transformer.up.b.{i}", (d, d, 3, 3))
            plain(f"u_transformer.up.b.{i + 1}", d)
        wrapped("fc", (self.out_chans * self.patch_height * self.patch_width, d))
        return spec


def named_fuxi_config(name: str) -> FuxiConfig:
    """Parity / bench geometries.  F6H = the model of BASELINE config 5 (the reference's fuxi_6h_single_step.yml: 0.25 degree, patch 4)."""
    if name == "FT0":    # rectangular patch, K padding in the embed GEMM, lat axis as tall as one window (no lat shift), lon padded
        return FuxiConfig(image_height=16, patch_height=2, image_width=48, patch_width=4, levels=2, frames=2, frame_patch_size=2, dim=64,
                          num_groups=(8, 16), channels=3, surface_channels=1, input_only_channels=0, output_only_channels=0, num_heads=2,
                          depth=2, window_size=4, meta_hidden=24, stage="cr")
    if name == "FT1":    # FuXi's window 7, both axes padded (9 -> 14, 11 -> 14), input-only channels, 4 channels per group as in FuXi
        return FuxiConfig(image_height=72, patch_height=4, image_width=88, patch_width=4, levels=2, frames=2, frame_patch_size=2, dim=128,
                          num_groups=32, channels=3, surface_channels=2, input_only_channels=1, output_only_channels=0, num_heads=2,
                          depth=3, window_size=7, meta_hidden=24, stage="cr")
    if name == "FT2":    # no spectral norm, single frame, output-only channel
        return FuxiConfig(image_height=32, patch_height=4, image_width=64, patch_width=4, levels=1, frames=1, frame_patch_size=1, dim=64,
                          num_groups=4, channels=4, surface_channels=3, input_only_channels=0, output_only_channels=1, num_heads=1,
                          depth=2, window_size=4, use_spectral_norm=False, meta_hidden=16, stage="cr")
    if name in ("FT0T", "FT1T", "FT2T"):   # the same three geometries with timm's block in the stage (the reference's own structure)
        import dataclasses
        return dataclasses.replace(named_fuxi_config(name[:-1]), stage="timm")
    if name == "F6H":    # config/gen_1/arXiv_2024/fuxi_6h_single_step.yml (model section): 74 channels in, 71 out, 266 M parameters
        return FuxiConfig(image_height=640, patch_height=4, image_width=1280, patch_width=4, levels=16, frames=2, frame_patch_size=2,
                          dim=1024, num_groups=32, channels=4, surface_channels=7, input_only_channels=3, output_only_channels=0,
                          num_heads=8, depth=16, window_size=7, stage="timm")
    if name == "F6HCR":  # the same with the V2-Cr stage (rounds 2-3 measured this one)
        import dataclasses
        return dataclasses.replace(named_fuxi_config("F6H"), stage="cr")
    raise KeyError(name)


def synth_fuxi_state_dict(cfg: FuxiConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Keyed synthetic weights (wxengine.synth): u / v after power iterations on weight_orig, as a trained checkpoint holds them."""
    spec = cfg.state_spec()
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in spec.items():
        if key.endswith((".weight_u", ".weight_v")):
            continue
        z = keyed_normal("fuxi/" + key, shape, seed)
        if key.endswith("logit_scale"):
            sd[key] = (np.log(10.0) + 0.3 * z).astype(np.float32)
        elif key.endswith((".q_bias", ".v_bias")):
            sd[key] = (0.1 * z).astype(np.float32)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            sd[key] = (z / np.sqrt(fan_in)).astype(np.float32)
        elif key.endswith(".weight"):
            sd[key] = (1.0 + 0.1 * z).astype(np.float32)
        else:
            sd[key] = (0.1 * z).astype(np.float32)
    for key in spec:
        if not key.endswith(".weight_u"):
            continue
        base = key[: -len(".weight_u")]
        w = sd[base + ".weight_orig"]
        w_mat = _sn_matrix(base, w)
        u, v = power_iterate(w_mat, keyed_normal("fuxi/" + key, (w_mat.shape[0],), seed))
        sd[base + ".weight_u"], sd[base + ".weight_v"] = u, v
    return OrderedDict((k, sd[k]) for k in spec)


def _sn_matrix(prefix: str, w: np.ndarray) -> np.ndarray:
    """torch.nn.utils.spectral_norm's matrix view: dim 0 first, except ConvTranspose2d (dim 1)."""
    if prefix.endswith("u_transformer.up.conv"):
        return np.moveaxis(w, 1, 0).reshape(w.shape[1], -1)
    return w.reshape(w.shape[0], -1)


# Modules whose weight the reference reads WITHOUT the spectral normalisation, timm stage only: timm's V2 WindowAttention never calls
# its `qkv` module -- it runs F.linear(x, self.qkv.weight, cat(q_bias, k_bias, v_bias)) -- so the forward pre-hook that
# torch.nn.utils.spectral_norm installs (fuxi.py:16-22) never fires for it, and the plain `weight` attribute it reads is the alias of
# `weight_orig` the hook-based implementation leaves behind (checked here with torch alone: after load_state_dict,
# `m.weight.data_ptr() == m.weight_orig.data_ptr()` until the module itself is called).  The effective qkv weight is weight_orig.
# PRECONDITION: that aliasing holds only for a module that has not been moved or cast since apply_spectral_norm (CPU, fp32, no
# `_apply`); FuxiConfig.timm_qkv_unnormalised = False selects the normalised weight for the other case.  No reference golden pins
# either choice (timm is not installable here): `tools/make_goldens.py --only fuxi_timm` generates one wherever `import timm` works.
TIMM_UNNORMALISED = (".attn.qkv",)


def fold_spectral_norm(sd, raw=()) -> "OrderedDict[str, np.ndarray]":
    """Eval-mode weights: every `<m>.weight_orig / weight_u / weight_v` triple becomes `<m>.weight = weight_orig / sigma`,
    sigma = u . (W_mat v) in fp32 as torch computes it (torch/nn/utils/spectral_norm.py compute_weight, do_power_iteration False).
    `raw`: module-name suffixes whose weight stays weight_orig (TIMM_UNNORMALISED)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    get = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32)  # noqa: E731
    for k in sd:
        if k.endswith(".weight_orig"):
            base = k[: -len(".weight_orig")]
            w = get(k)
            if raw and base.endswith(tuple(raw)):
                out[base + ".weight"] = w
                continue
            sigma = np.float32(np.dot(get(base + ".weight_u"), _sn_matrix(base, w) @ get(base + ".weight_v")))
            out[base + ".weight"] = (w / sigma).astype(np.float32)
        elif k.endswith((".weight_u", ".weight_v")):
            continue
        else:
            out[k] = get(k)
    return out


class FuxiHIP:
    """Fuxi.forward (fuxi.py:454-500) on MI355X: x [B, C_in, frames, H, W] float32 on the GPU -> [B, C_out, 1, H, W] float32."""

    def __init__(self, precision: str = "bf16", device: Optional[int] = None, **model_conf):
        import torch
        if not torch.cuda.is_available():
            raise WXEngineError("no GPU visible: the FuXi engine has no CPU fallback")
        self.cfg = cfg = model_conf.pop("cfg", None) or FuxiConfig.from_model_conf(model_conf)
        cfg.check()
        self.lib = load_library()
        self.precision = precision
        self.device = torch.cuda.current_device() if device is None else int(device)
        g = cfg.groups
        d = wx_fuxi_desc(PREC[precision], cfg.image_height, cfg.image_width, cfg.in_chans, cfg.out_chans, cfg.frames, cfg.patch_height,
                         cfg.patch_width, cfg.dim, cfg.num_heads, cfg.window_size, cfg.depth, g[0], g[1], STAGE_VARIANT[cfg.stage])
        self._h = C.c_void_p()
        self.lib.wx_fuxi_create.argtypes = [C.POINTER(wx_fuxi_desc), C.c_int, C.POINTER(C.c_void_p)]
        _check(self.lib.wx_fuxi_create(C.byref(d), self.device, C.byref(self._h)))
        self._loaded = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.wx_fuxi_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _put(self, name: str, arr) -> None:
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        _check(self.lib.wx_fuxi_load(self._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(a.size)))

    def load_state_dict(self, sd, strict: bool = True) -> None:
        """sd with the reference's keys (see the module docstring); strict: unknown / missing keys raise KeyError."""
        cfg = self.cfg
        spec = cfg.state_spec()
        missing = [k for k in spec if k not in sd]
        extra = [k for k in sd if k not in spec and not k.endswith(TIMM_DERIVED_SUFFIXES)]
        if missing or (strict and extra):
            raise KeyError(f"FuxiHIP.load_state_dict: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                           f"unexpected {extra[:4]}{'...' if len(extra) > 4 else ''}")
        for k, shape in spec.items():
            got = tuple(sd[k].shape)
            if got != tuple(shape):
                raise ValueError(f"FuxiHIP.load_state_dict: {k} has shape {got}, expected {tuple(shape)}")
        eff = fold_spectral_norm(OrderedDict((k, sd[k]) for k in spec),
                                 raw=TIMM_UNNORMALISED if (cfg.stage == "timm" and cfg.timm_qkv_unnormalised) else ())
        ws = (cfg.window_size, cfg.window_size)
        if cfg.stage == "timm":
            for k, v in eff.items():
                if not k.startswith("u_transformer.layer.blocks."):
                    self._put(k, v)
            for i in range(cfg.depth):
                p = f"u_transformer.layer.blocks.{i}."
                for name, arr in timm_block_tensors(lambda k: eff[k], p, ws, cfg.dim).items():
                    self._put(p + name, arr)
            _check(self.lib.wx_fuxi_finalize(self._h))
            self._loaded = True
            return
        for k, v in eff.items():
            if ".attn.meta_mlp." in k or k.endswith("attn.logit_scale"):
                continue
            self._put(k, v)
        for i in range(cfg.depth):
            p = f"u_transformer.layer.blocks.{i}.attn."
            self._put(p + "bias_table", relative_position_bias(eff[p + "meta_mlp.fc1.weight"], eff[p + "meta_mlp.fc1.bias"],
                                                               eff[p + "meta_mlp.fc2.weight"], eff[p + "meta_mlp.fc2.bias"], ws))
            self._put(p + "logit_scale", effective_logit_scale(eff[p + "logit_scale"]))
        _check(self.lib.wx_fuxi_finalize(self._h))
        self._loaded = True

    @property
    def flops(self) -> float:
        f = C.c_double()
        _check(self.lib.wx_fuxi_flops(self._h, C.byref(f)))
        return float(f.value)

    def debug_map(self, name: str) -> np.ndarray:
        shape = (C.c_int64 * 3)()
        _check(self.lib.wx_fuxi_debug_map(self._h, name.encode(), None, C.c_int64(0), shape))
        out = np.empty(tuple(int(s) for s in shape), dtype=np.float32)
        _check(self.lib.wx_fuxi_debug_map(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(out.size), shape))
        return out

    def forward(self, x, out=None):
        import torch
        cfg = self.cfg
        if not self._loaded:
            raise WXEngineError("load_state_dict first")
        want = (cfg.in_chans, cfg.frames, cfg.image_height, cfg.image_width)
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and tuple(x.shape[1:]) == want):
            raise WXEngineError(f"x must be a float32 GPU tensor [B, {', '.join(map(str, want))}], got "
                                f"{tuple(x.shape) if hasattr(x, 'shape') else type(x)}")
        if x.device.index != self.device:
            raise WXEngineError(f"x is on cuda:{x.device.index}, the model was created for cuda:{self.device}")
        x = x.contiguous()
        B = x.shape[0]
        if out is None:
            out = torch.empty((B, cfg.out_chans, 1, cfg.image_height, cfg.image_width), dtype=torch.float32, device=x.device)
        with torch.cuda.device(self.device):
            s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            for b in range(B):
                _check(self.lib.wx_fuxi_forward(self._h, C.c_void_p(x[b].data_ptr()), C.c_void_p(out[b].data_ptr()), s))
        return out

    __call__ = forward
