"""Window attention as an operator of its own: the Swin / FuXi mode of the engine's attention kernel.

`WindowAttention` is the attention CORE of a windowed transformer block -- everything between the qkv projection and the output
projection of credit/models/swin.py::WindowMultiHeadAttention.forward (:299-330) inside `_shifted_window_attn` (:451-486): cyclic
shift, window partition, scaled cosine (or dot-product) scores + relative position bias + seam mask, softmax, P V, merge, shift
back -- on a token-major map resident in HBM, through the C ABI (`wx_winattn_*`).  The FuXi stage (credit/models/fuxi.py:250-260)
runs the same operator through timm's SwinTransformerV2Stage.  There is no CPU fallback.

`relative_position_bias` / `effective_logit_scale` turn a reference checkpoint's `attn.meta_mlp.*` / `attn.logit_scale` tensors into
the operator's inputs (swin.py:254-297, :307); they run once at load time on the host.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence, Tuple

import numpy as np

from .engine import PREC, WXEngineError, _check, load_library

KIND = {"block": 0, "dilated": 1, "shifted": 3}


class wx_winattn_desc(C.Structure):
    _fields_ = [("precision", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32),
                ("head_dim", C.c_int32), ("wsz_y", C.c_int32), ("wsz_x", C.c_int32), ("kind", C.c_int32), ("shift_y", C.c_int32),
                ("shift_x", C.c_int32), ("softmax_scale", C.c_float), ("mask_value", C.c_float), ("mask_axes", C.c_int32)]


def cpb_position_bias(w0, b0, w2, window: Tuple[int, int]) -> np.ndarray:
    """timm's Swin V2 continuous position bias, [heads, N, N]: `16 * sigmoid(cpb_mlp(relative_coords_table))[relative_position_index]`
    (timm.models.swin_transformer_v2.WindowAttention: cpb_mlp = Linear(2, 512) -> ReLU -> Linear(512, heads, bias=False); the table
    holds the offsets -(w-1)..(w-1) per axis, divided by (w-1), times 8, then sign(x) * log2(|x| + 1) / log2(8)).  Host, float64."""
    wh, ww = window
    dy = np.arange(-(wh - 1), wh, dtype=np.float64) / max(wh - 1, 1)
    dx = np.arange(-(ww - 1), ww, dtype=np.float64) / max(ww - 1, 1)
    tab = np.stack(np.meshgrid(dy, dx, indexing="ij"), axis=-1) * 8.0                       # [2wh-1, 2ww-1, 2]
    tab = np.sign(tab) * np.log2(np.abs(tab) + 1.0) / np.log2(8.0)
    h = np.maximum(tab.reshape(-1, 2) @ np.asarray(w0, np.float64).T + np.asarray(b0, np.float64), 0.0)
    t = 16.0 / (1.0 + np.exp(-(h @ np.asarray(w2, np.float64).T)))                           # [(2wh-1)(2ww-1), heads]
    ys, xs = np.meshgrid(np.arange(wh), np.arange(ww), indexing="ij")
    cy, cx = ys.ravel(), xs.ravel()
    idx = (cy[:, None] - cy[None, :] + wh - 1) * (2 * ww - 1) + (cx[:, None] - cx[None, :] + ww - 1)   # [N, N], query-major
    return np.ascontiguousarray(t[idx.ravel()].reshape(wh * ww, wh * ww, -1).transpose(2, 0, 1), dtype=np.float32)


def relative_position_bias(fc1_w, fc1_b, fc2_w, fc2_b, window: Tuple[int, int]) -> np.ndarray:
    """[heads, N, N] from the meta network's weights (swin.py:254-297: log-spaced relative coordinates -> Linear, ReLU, Linear)."""
    ys, xs = np.meshgrid(np.arange(window[0]), np.arange(window[1]), indexing="ij")
    coords = np.stack([ys.ravel(), xs.ravel()]).astype(np.float64)                 # [2, N]
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).reshape(-1, 2)
    rel = np.sign(rel) * np.log1p(np.abs(rel))
    h = np.maximum(rel @ np.asarray(fc1_w, np.float64).T + np.asarray(fc1_b, np.float64), 0.0)
    t = h @ np.asarray(fc2_w, np.float64).T + np.asarray(fc2_b, np.float64)         # [N*N, heads]
    n = window[0] * window[1]
    return np.ascontiguousarray(t.T.reshape(-1, n, n), dtype=np.float32)


def effective_logit_scale(raw) -> np.ndarray:
    """swin.py:307: exp(clamp(logit_scale, max = log(1 / 0.01)))."""
    return np.exp(np.minimum(np.asarray(raw, np.float64), math.log(1.0 / 0.01))).astype(np.float32)


class WindowAttention:
    def __init__(self, feat: Tuple[int, int], heads: int, head_dim: int, window, shift: Sequence[int] = (0, 0), kind: Optional[str] = None,
                 bias=None, logit_scale=None, softmax_scale: Optional[float] = None, mask_value: float = -100.0,
                 precision: str = "bf16", device: Optional[int] = None, mask_axes: int = 1):
        """bias: [heads, N, N] or [1, N, N] or None; logit_scale: [heads] (already exponentiated) selects cosine attention;
        mask_axes: 1 = the latitude-only seam mask of swin.py:411-427, 3 = both axes (timm's SwinTransformerV2Block)."""
        import torch
        if not torch.cuda.is_available():
            raise WXEngineError("no GPU visible: window attention has no CPU fallback")
        self.lib = load_library()
        ws = (int(window), int(window)) if np.isscalar(window) else (int(window[0]), int(window[1]))
        self.feat, self.heads, self.head_dim, self.window, self.shift = tuple(feat), heads, head_dim, ws, (int(shift[0]), int(shift[1]))
        self.precision = precision
        self.device = torch.cuda.current_device() if device is None else int(device)
        if kind is None:
            kind = "shifted" if any(self.shift) else "block"
        d = wx_winattn_desc(PREC[precision], feat[0], feat[1], heads * head_dim, heads, head_dim, ws[0], ws[1], KIND[kind],
                            self.shift[0], self.shift[1], float(softmax_scale if softmax_scale is not None else head_dim ** -0.5),
                            float(mask_value), int(mask_axes))
        fp = C.POINTER(C.c_float)
        n = ws[0] * ws[1]
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32).reshape(-1, n, n)
        ls = None if logit_scale is None else np.ascontiguousarray(logit_scale, dtype=np.float32).ravel()
        if ls is not None and ls.size != heads:
            raise ValueError("logit_scale needs one value per head")
        self._h = C.c_void_p()
        self.lib.wx_winattn_create.argtypes = [C.POINTER(wx_winattn_desc), fp, C.c_int, fp, C.c_int, C.POINTER(C.c_void_p)]
        _check(self.lib.wx_winattn_create(C.byref(d), None if b is None else b.ctypes.data_as(fp), 0 if b is None else b.shape[0],
                                          None if ls is None else ls.ctypes.data_as(fp), self.device, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.wx_winattn_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def __call__(self, qkv, out=None):
        """qkv [H, W, 3C] (or [H*W, 3C]) on the GPU, bf16 / float32 matching `precision`, q | k | v head-major -> [H, W, C]."""
        import torch
        want = torch.bfloat16 if self.precision == "bf16" else torch.float32
        H, W = self.feat
        c = self.heads * self.head_dim
        if not (isinstance(qkv, torch.Tensor) and qkv.is_cuda and qkv.dtype == want and qkv.is_contiguous()):
            raise WXEngineError(f"qkv must be a contiguous {want} tensor on the GPU")
        if qkv.numel() != H * W * 3 * c or qkv.shape[-1] != 3 * c:
            raise WXEngineError(f"qkv has shape {tuple(qkv.shape)}, expected [{H}, {W}, {3 * c}]")
        if qkv.device.index != self.device:
            raise WXEngineError(f"qkv is on cuda:{qkv.device.index}, the operator was created for cuda:{self.device}")
        if out is None:
            out = torch.empty((H, W, c), dtype=want, device=qkv.device)
        elif not (out.is_cuda and out.dtype == want and out.is_contiguous() and out.numel() == H * W * c):
            raise WXEngineError("out must be a contiguous tensor of H*W*C elements in the operator's precision")
        with torch.cuda.device(self.device):
            _check(self.lib.wx_winattn_apply(self._h, C.c_void_p(qkv.data_ptr()), C.c_void_p(out.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out


class wx_swin_desc(C.Structure):
    _fields_ = [("precision", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32),
                ("wsz_y", C.c_int32), ("wsz_x", C.c_int32), ("depth", C.c_int32), ("hidden", C.c_int32), ("shift_y", C.c_int32),
                ("shift_x", C.c_int32), ("mask_value", C.c_float), ("ln_eps", C.c_float), ("mask_axes", C.c_int32)]


# keys of timm's stage that carry no information the engine needs: non-persistent buffers in timm >= 0.9 (older releases saved them)
TIMM_DERIVED_SUFFIXES = ("attn.relative_coords_table", "attn.relative_position_index", "attn.k_bias", "attn_mask")


def timm_block_tensors(get, p: str, window: Tuple[int, int], dim: int):
    """One block of timm's SwinTransformerV2Stage (EFFECTIVE weights under prefix `p`) -> the engine's wx_swin_load tensors:
    qkv bias = (q_bias | 0 | v_bias) (the k bias is a zero buffer), bias table = cpb_position_bias, logit scale = exp(clamp(., log 100))."""
    out = {name: get(p + name) for name in ("attn.qkv.weight", "attn.proj.weight", "attn.proj.bias", "norm1.weight", "norm1.bias",
                                            "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "norm2.weight", "norm2.bias")}
    out["attn.qkv.bias"] = np.concatenate([get(p + "attn.q_bias").ravel(), np.zeros(dim, np.float32), get(p + "attn.v_bias").ravel()])
    out["attn.bias_table"] = cpb_position_bias(get(p + "attn.cpb_mlp.0.weight"), get(p + "attn.cpb_mlp.0.bias"), get(p + "attn.cpb_mlp.2.weight"), window)
    out["attn.logit_scale"] = effective_logit_scale(get(p + "attn.logit_scale").ravel())
    return out


class SwinStage:
    """`depth` Swin V2 (Cr) blocks on a token-major map resident in HBM (C ABI `wx_swin_*`): the host-side mirror of
    credit/models/swin.py::SwinTransformerV2CrStage (:560-668, downscale = False) made of SwinTransformerV2CrBlock (:330-502).
    Same constructor vocabulary (`dim`, `num_heads`, `feat_size`, `window_size`, `mlp_ratio`), same state-dict keys
    (`blocks.{i}.attn.qkv.weight`, `blocks.{i}.attn.meta_mlp.fc1.weight`, `blocks.{i}.attn.logit_scale`, `blocks.{i}.norm1.weight`,
    `blocks.{i}.mlp.fc1.weight`, ...): a reference checkpoint's stage loads unchanged; the meta MLP of every block is evaluated
    once, on the host, into that block's [heads, N, N] bias table.  Even blocks are unshifted, odd blocks shifted by window // 2
    (a window as large as the map is not shifted, swin.py:407-409).  No CPU fallback."""

    _DIRECT = ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm1.weight", "norm1.bias",
               "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "norm2.weight", "norm2.bias")

    def __init__(self, dim: int, depth: int, num_heads: int, feat_size: Tuple[int, int], window_size, mlp_ratio: float = 4.0,
                 precision: str = "bf16", device: Optional[int] = None, variant: str = "cr"):
        """variant "cr": credit/models/swin.py's SwinTransformerV2CrBlock (keys attn.meta_mlp.*, attn.qkv.bias; latitude-only mask).
        variant "timm": timm.models.swin_transformer_v2.SwinTransformerV2Block, the block FuXi's stage is made of (fuxi.py:250-260):
        keys attn.q_bias / attn.v_bias / attn.cpb_mlp.{0,2}.* / attn.logit_scale [heads, 1, 1], mask over both axes."""
        import torch
        if variant not in ("cr", "timm"):
            raise ValueError("SwinStage: variant must be 'cr' or 'timm'")
        self.variant = variant
        if not torch.cuda.is_available():
            raise WXEngineError("no GPU visible: the Swin stage has no CPU fallback")
        self.lib = load_library()
        ws = (int(window_size), int(window_size)) if np.isscalar(window_size) else (int(window_size[0]), int(window_size[1]))
        # swin.py:405-409 `_calc_window_shift`: a window is clipped to the map, and a clipped axis is not shifted
        self.window = tuple(f if f <= w else w for f, w in zip(feat_size, ws))
        self.shift = tuple(0 if f <= w else w // 2 for f, w in zip(feat_size, self.window))
        self.dim, self.depth, self.heads, self.feat = int(dim), int(depth), int(num_heads), (int(feat_size[0]), int(feat_size[1]))
        self.hidden = int(dim * mlp_ratio)
        self.precision = precision
        self.device = torch.cuda.current_device() if device is None else int(device)
        d = wx_swin_desc(PREC[precision], self.feat[0], self.feat[1], self.dim, self.heads, self.window[0], self.window[1], self.depth,
                         self.hidden, self.shift[0], self.shift[1], -100.0, 1e-5, 3 if variant == "timm" else 1)
        self._h = C.c_void_p()
        self.lib.wx_swin_create.argtypes = [C.POINTER(wx_swin_desc), C.c_int, C.POINTER(C.c_void_p)]
        _check(self.lib.wx_swin_create(C.byref(d), self.device, C.byref(self._h)))
        self._loaded = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.wx_swin_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _put(self, block: int, name: str, arr) -> None:
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        _check(self.lib.wx_swin_load(self._h, block, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size))

    def load_state_dict(self, sd, prefix: str = "blocks.") -> None:
        """sd: {key: array-like} with the reference stage's keys; missing keys raise (KeyError names the first one)."""
        get = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32)  # noqa: E731
        for i in range(self.depth):
            p = f"{prefix}{i}."
            if self.variant == "timm":
                for name, arr in timm_block_tensors(get, p, self.window, self.dim).items():
                    self._put(i, name, arr)
                continue
            for name in self._DIRECT:
                self._put(i, name, get(p + name))
            self._put(i, "attn.bias_table", relative_position_bias(get(p + "attn.meta_mlp.fc1.weight"), get(p + "attn.meta_mlp.fc1.bias"),
                                                                  get(p + "attn.meta_mlp.fc2.weight"), get(p + "attn.meta_mlp.fc2.bias"),
                                                                  self.window))
            self._put(i, "attn.logit_scale", effective_logit_scale(get(p + "attn.logit_scale")))
        _check(self.lib.wx_swin_finalize(self._h))
        self._loaded = True

    @property
    def flops(self) -> float:
        f = C.c_double()
        _check(self.lib.wx_swin_flops(self._h, C.byref(f)))
        return float(f.value)

    def __call__(self, x, out=None):
        """x [H, W, C] (or [H*W, C]) on the GPU in the stage's precision -> the stage output, same shape (out may be x)."""
        import torch
        want = torch.bfloat16 if self.precision == "bf16" else torch.float32
        H, W = self.feat
        if not self._loaded:
            raise WXEngineError("load_state_dict first")
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == want and x.is_contiguous() and x.numel() == H * W * self.dim
                and x.shape[-1] == self.dim):
            raise WXEngineError(f"x must be a contiguous {want} tensor [{H}, {W}, {self.dim}] on the GPU")
        if x.device.index != self.device:
            raise WXEngineError(f"x is on cuda:{x.device.index}, the stage was created for cuda:{self.device}")
        if out is None:
            out = torch.empty_like(x)
        elif not (out.is_cuda and out.dtype == want and out.is_contiguous() and out.shape == x.shape):
            raise WXEngineError("out must match x")
        with torch.cuda.device(self.device):
            _check(self.lib.wx_swin_apply(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out


class Attend:
    """credit/attend.py::Attend (:40-120) for inference on the GPU: out = softmax(q k^T * scale) v per batch item and head, the
    non-windowed "thin mode" of the attention operator -- every batch item is one window of the operator's
    token map.  Same constructor vocabulary (`dropout` must be 0, `flash` is accepted and ignored: the HIP kernel IS the fused path,
    `scale` None = head_dim ** -0.5).  q, k, v [b, h, n, d] on the GPU, n <= 128 tokens, d in {32, 64, 96, 128}; no CPU fallback."""

    def __init__(self, dropout: float = 0.0, flash: bool = False, scale: Optional[float] = None, precision: str = "bf16"):
        if dropout:
            raise ValueError("Attend (HIP): dropout is a training-time option; the engine runs the eval forward")
        self.scale, self.flash, self.precision = scale, flash, precision
        self._ops = {}

    @staticmethod
    def _window(n: int) -> Tuple[int, int]:
        wx = max(d for d in range(1, 17) if n % d == 0)
        return n // wx, wx

    def __call__(self, q, k, v):
        import torch
        if not (q.is_cuda and q.dim() == 4 and q.shape == k.shape == v.shape):
            raise WXEngineError("Attend: q, k, v must be GPU tensors of one shape [b, h, n, d]")
        b, h, n, d = q.shape
        if n > 128:
            raise WXEngineError("Attend (HIP): at most 128 tokens per sequence (one window of the attention kernel)")
        wy, wx = self._window(n)
        dt = torch.bfloat16 if self.precision == "bf16" else torch.float32
        key = (b, h, n, d, q.device.index)
        if key not in self._ops:
            self._ops[key] = WindowAttention((b * wy, wx), h, d, (wy, wx), (0, 0), kind="block", bias=None,
                                             softmax_scale=self.scale if self.scale is not None else d ** -0.5,
                                             precision=self.precision, device=q.device.index)
        # [b, h, n, d] x 3 -> the operator's token-major map [b * wy, wx, 3 h d] (q | k | v, head-major)
        qkv = torch.stack((q, k, v), 0).permute(1, 3, 0, 2, 4).reshape(b * wy, wx, 3 * h * d).to(dt).contiguous()
        o = self._ops[key](qkv)                                             # [b * wy, wx, h d]
        return o.reshape(b, n, h, d).permute(0, 2, 1, 3).to(q.dtype)

    forward = __call__
