"""CPU oracle for the second architecture's hot op: shifted-window attention of the Swin V2 (Cr) block.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product package (miles-credit_amd/) or bench.py's timed region.
Pinned: tests/test_swin_oracle.py checks it against tests/golden/swin_*.npz, which tools/make_goldens.py writes by running the
reference's own `SwinTransformerV2CrBlock` (credit/models/swin.py) in the dev container.

Restates, in plain torch on the CPU:
  relative_position_bias   swin.py:254-297  log-spaced relative coordinates -> meta MLP (Linear, ReLU, Linear) -> [heads, N, N]
  shift_mask               swin.py:411-427  two latitude regions of the rolled map, -100 between them (longitude is periodic)
  window_attention_core    swin.py:299-330 + :451-486  roll, partition, cosine / dot attention + bias + mask, softmax, P V, merge,
                           roll back -- everything between the qkv projection and the output projection
  block                    swin.py:488-505  x + norm1(proj(core(qkv(x)))), then x + norm2(mlp(x))
  attend                   credit/attend.py:94-120  the non-windowed softmax(q k^T scale) v (pinned by tests/golden/attend.npz)
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


def relative_coordinates_log(ws: Tuple[int, int], dtype=torch.float32) -> Tensor:
    """swin.py:254-270: pairwise (dy, dx) between the window's tokens, sign(d) * log(1 + |d|); [N*N, 2], query-major."""
    ys, xs = torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), indexing="ij")
    coords = torch.stack([ys, xs]).flatten(1)                        # [2, N]
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).reshape(-1, 2).to(dtype)
    return torch.sign(rel) * torch.log1p(rel.abs())


def relative_position_bias(sd: dict, prefix: str, ws: Tuple[int, int], heads: int, dtype=torch.float32) -> Tensor:
    """swin.py:283-297 with the meta MLP of :241-252 in eval mode (dropout off): [heads, N, N]."""
    t = relative_coordinates_log(ws, dtype)
    t = F.relu(F.linear(t, sd[prefix + "meta_mlp.fc1.weight"].to(dtype), sd[prefix + "meta_mlp.fc1.bias"].to(dtype)))
    t = F.linear(t, sd[prefix + "meta_mlp.fc2.weight"].to(dtype), sd[prefix + "meta_mlp.fc2.bias"].to(dtype))
    n = ws[0] * ws[1]
    return t.transpose(1, 0).reshape(heads, n, n)


def effective_logit_scale(raw: Tensor) -> Tensor:
    """swin.py:307: exp(clamp(logit_scale, max = log(1 / 0.01)))."""
    return torch.clamp(raw, max=math.log(1.0 / 0.01)).exp()


def shift_mask(feat: Tuple[int, int], ws: Tuple[int, int], shift: Tuple[int, int], dtype=torch.float32) -> Optional[Tensor]:
    """swin.py:411-427: [num_windows, N, N] of 0 / -100, or None without a shift.  Only latitude rows are split into regions."""
    if not any(shift):
        return None
    H, W = feat
    img = torch.zeros(H, W, dtype=dtype)
    img[H - shift[0]:] = 1.0 if shift[0] else 0.0     # rows [0, H - ws) and [H - ws, H - shift) carry 0, the last `shift` rows 1
    win = img.reshape(H // ws[0], ws[0], W // ws[1], ws[1]).permute(0, 2, 1, 3).reshape(-1, ws[0] * ws[1])
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def window_attention_core(qkv: Tensor, heads: int, ws: Tuple[int, int], shift: Tuple[int, int], bias: Optional[Tensor],
                          logit_scale: Optional[Tensor] = None, softmax_scale: Optional[float] = None) -> Tensor:
    """qkv [H, W, 3C] (q | k | v, head-major) -> [H, W, C].  logit_scale ([heads], already exponentiated) selects cosine
    attention; otherwise scores = q k^T * softmax_scale (default 1/sqrt(head_dim))."""
    H, W, c3 = qkv.shape
    C = c3 // 3
    hd = C // heads
    n = ws[0] * ws[1]
    x = torch.roll(qkv, shifts=(-shift[0], -shift[1]), dims=(0, 1)) if any(shift) else qkv
    win = x.reshape(H // ws[0], ws[0], W // ws[1], ws[1], 3, heads, hd).permute(4, 0, 2, 5, 1, 3, 6).reshape(3, -1, heads, n, hd)
    q, k, v = win[0], win[1], win[2]                                    # [num_windows, heads, N, hd]
    if logit_scale is not None:
        attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1) * logit_scale.reshape(1, heads, 1, 1)
    else:
        attn = q @ k.transpose(-2, -1) * (softmax_scale if softmax_scale is not None else hd ** -0.5)
    if bias is not None:
        attn = attn + bias.unsqueeze(0)
    m = shift_mask((H, W), ws, shift, qkv.dtype)
    if m is not None:
        attn = attn + m.unsqueeze(1)
    out = attn.softmax(dim=-1) @ v                                      # [num_windows, heads, N, hd]
    out = out.reshape(H // ws[0], W // ws[1], heads, ws[0], ws[1], hd).permute(0, 3, 1, 4, 2, 5).reshape(H, W, C)
    return torch.roll(out, shifts=shift, dims=(0, 1)) if any(shift) else out


def block(x: Tensor, sd: dict, heads: int, ws: Tuple[int, int], shift: Tuple[int, int], prefix: str = "") -> Tensor:
    """SwinTransformerV2CrBlock.forward (swin.py:488-505), post-norm residual branches; x [H, W, C]."""
    H, W, C = x.shape
    qkv = F.linear(x, sd[prefix + "attn.qkv.weight"], sd[prefix + "attn.qkv.bias"])
    core = window_attention_core(qkv, heads, ws, shift, relative_position_bias(sd, prefix + "attn.", ws, heads, x.dtype),
                                 effective_logit_scale(sd[prefix + "attn.logit_scale"]))
    a = F.linear(core, sd[prefix + "attn.proj.weight"], sd[prefix + "attn.proj.bias"])
    x = x + F.layer_norm(a, (C,), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], 1e-5)
    h = F.gelu(F.linear(x, sd[prefix + "mlp.fc1.weight"], sd[prefix + "mlp.fc1.bias"]))
    h = F.linear(h, sd[prefix + "mlp.fc2.weight"], sd[prefix + "mlp.fc2.bias"])
    return x + F.layer_norm(h, (C,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], 1e-5)


def attend(q: Tensor, k: Tensor, v: Tensor, scale: Optional[float] = None) -> Tensor:
    """credit/attend.py:94-120 (Attend.forward, non-flash branch): softmax(q k^T * scale) v, q / k / v [b, h, n, d]."""
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (scale if scale is not None else q.shape[-1] ** -0.5)
    return torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
