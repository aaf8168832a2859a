"""CPU oracle for BASELINE config 5: the FuXi forward around a Swin V2 (Cr) stage.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product package).  Pinned: tests/test_fuxi.py checks it against
tests/golden/fuxi_*.npz, which tools/make_goldens.py::fuxi_golden writes by running the reference's own modules in the dev
container -- `CubeEmbedding`, `DownBlock`, `UpBlock`, `UTransformer.forward`, `Fuxi.forward`, `apply_spectral_norm`, `get_pad2d`
of credit/models/fuxi.py, with `SwinTransformerV2CrBlock.forward` of credit/models/swin.py in the stage's place (the reference
instantiates timm's stage there; timm is not vendored -- that substitution is the unpinned part).

Restates, in plain torch on the CPU (fp32 or fp64 by the dtype of x):
  effective_weights   torch.nn.utils.spectral_norm in eval mode: W = weight_orig / (u . W_mat v); dim 1 for ConvTranspose2d
  cube_embedding      fuxi.py:124-143  Conv3d(kernel = stride = patch) -> LayerNorm over channels
  down_block          fuxi.py:162-173  conv3x3 s2 -> [conv3x3, GroupNorm, SiLU] x 2 + shortcut
  u_transformer       fuxi.py:278-310  down, zero-pad to the window (:31-65), stage, crop, concat [shortcut, x], up
  up_block            fuxi.py:191-201  ConvTranspose2d k2 s2 -> [conv3x3, GroupNorm, SiLU] x 2 + shortcut
  forward             fuxi.py:454-500  embed, u_transformer, fc on channels-last, patch -> pixel reshape
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from oracle import swin_oracle as S


def effective_weights(sd: Dict[str, Tensor], dtype=torch.float32, raw: Tuple[str, ...] = ()) -> Dict[str, Tensor]:
    """raw: module suffixes read without the normalisation -- timm's WindowAttention calls F.linear(x, self.qkv.weight, ...) and never
    the qkv module, so spectral_norm's forward pre-hook does not fire and `weight` stays the alias of weight_orig (variant "timm")."""
    out = {}
    for k, v in sd.items():
        v = torch.as_tensor(v)
        if k.endswith(".weight_orig"):
            base = k[: -len(".weight_orig")]
            w = v.to(torch.float32)
            if raw and base.endswith(tuple(raw)):
                out[base + ".weight"] = w.to(dtype)
                continue
            mat = w.transpose(0, 1).reshape(w.shape[1], -1) if base.endswith("u_transformer.up.conv") else w.reshape(w.shape[0], -1)
            sigma = torch.dot(torch.as_tensor(sd[base + ".weight_u"]).float(), mat @ torch.as_tensor(sd[base + ".weight_v"]).float())
            out[base + ".weight"] = (w / sigma).to(dtype)
        elif not k.endswith((".weight_u", ".weight_v")):
            out[k] = v.to(dtype)
    return out


def window_padding(n: int, window: int) -> Tuple[int, int]:
    rem = n % window
    tot = window - rem if rem else 0
    return tot // 2, tot - tot // 2


def cube_embedding(x: Tensor, w: Dict[str, Tensor]) -> Tensor:
    """x [C, T, H, W] -> [dim, H / ph, W / pw]."""
    pw = w["cube_embedding.proj.weight"]
    y = F.conv3d(x[None], pw, w["cube_embedding.proj.bias"], stride=pw.shape[2:])[0]          # [dim, 1, Hp, Wp]
    d, _, hp, wp = y.shape
    y = F.layer_norm(y.reshape(d, -1).T, (d,), w["cube_embedding.norm.weight"], w["cube_embedding.norm.bias"], 1e-5)
    return y.T.reshape(d, hp, wp)


def _residual_path(x: Tensor, w: Dict[str, Tensor], p: str, groups: int) -> Tensor:
    sc = x
    for i in (0, 3):
        x = F.conv2d(x[None], w[f"{p}b.{i}.weight"], w[f"{p}b.{i}.bias"], padding=1)
        x = F.silu(F.group_norm(x, groups, w[f"{p}b.{i + 1}.weight"], w[f"{p}b.{i + 1}.bias"], 1e-5))[0]
    return x + sc


def down_block(x: Tensor, w: Dict[str, Tensor], groups: int) -> Tensor:
    p = "u_transformer.down."
    return _residual_path(F.conv2d(x[None], w[p + "conv.weight"], w[p + "conv.bias"], stride=2, padding=1)[0], w, p, groups)


def up_block(x: Tensor, w: Dict[str, Tensor], groups: int) -> Tensor:
    p = "u_transformer.up."
    return _residual_path(F.conv_transpose2d(x[None], w[p + "conv.weight"], w[p + "conv.bias"], stride=2)[0], w, p, groups)


def stage(x: Tensor, w: Dict[str, Tensor], heads: int, window: int, depth: int, prefix: str = "u_transformer.layer.blocks.",
          variant: str = "cr") -> Tensor:
    """x [H, W, C]; blocks alternate shift 0 / window // 2, an axis as large as its window is not shifted (swin.py:405-409).
    variant "timm": timm's Swin V2 block (what fuxi.py:250-260 instantiates; swin_oracle.stage_timm, parity unpinned)."""
    if variant == "timm":
        return S.stage_timm(x, w, heads, window, depth, prefix=prefix)
    H, W, _ = x.shape
    ws = (min(window, H), min(window, W))
    for i in range(depth):
        shift = (0, 0) if i % 2 == 0 else tuple(0 if f <= s else s // 2 for f, s in zip((H, W), ws))
        x = S.block(x, w, heads, ws, shift, prefix=f"{prefix}{i}.")
    return x


def forward(x: Tensor, sd: Dict[str, Tensor], heads: int, window: int, depth: int, groups: Tuple[int, int], out_chans: int,
            taps: dict = None, variant: str = "cr") -> Tensor:
    """x [C_in, T, H, W] -> y [C_out, H, W]; `taps` (optional dict) receives the intermediate maps, channels-last."""
    w = effective_weights(sd, x.dtype, raw=(".attn.qkv",) if variant == "timm" else ())
    ph, pw = w["cube_embedding.proj.weight"].shape[3:]
    e = cube_embedding(x, w)
    d = down_block(e, w, groups[0])
    _, hd, wd = d.shape
    (pt, pb), (pl, pr) = window_padding(hd, window), window_padding(wd, window)
    t = stage(F.pad(d, (pl, pr, pt, pb)).permute(1, 2, 0), w, heads, window, depth, variant=variant).permute(2, 0, 1)
    t = t[:, pt: pt + hd, pl: pl + wd]
    u = up_block(torch.cat([d, t], 0), w, groups[1])
    f = F.linear(u.permute(1, 2, 0), w["fc.weight"], w["fc.bias"])                               # [Hp, Wp, ph pw C_out]
    hp, wp, _ = f.shape
    y = f.reshape(hp, wp, ph, pw, out_chans).permute(0, 2, 1, 3, 4).reshape(hp * ph, wp * pw, out_chans).permute(2, 0, 1)
    if taps is not None:
        taps.update(embed=e.permute(1, 2, 0), down=d.permute(1, 2, 0), stage=t.permute(1, 2, 0), up=u.permute(1, 2, 0))
    return y
