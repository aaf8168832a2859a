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

PARITY UNPINNED (timm absent): `cpb_position_bias`, `shift_mask_timm`, `block_timm`, `stage_timm` restate the PUBLISHED block of
timm.models.swin_transformer_v2 (SwinTransformerV2Block / WindowAttention / SwinTransformerV2Stage, timm 0.9.x - 1.0.x) -- the class
credit/models/fuxi.py:4-5, 250-260 instantiates.  timm is neither vendored nor pinned by the reference and cannot be installed here
(SURVEY.md 8(c)), so no golden of the real class exists; what anchors these functions: the reference's call site (constructor
arguments, BHWC in / out, fuxi.py:250-260, 285-287), timm's state-dict key names and shapes, and everything they share with the
pinned V2-Cr functions above (roll / partition / merge, cosine attention, clamped logit scale, res-post-norm).
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


# ---- timm's Swin V2 block (parity unpinned, see the header) ------------------------------------------------------------------------
def cpb_position_bias(sd: dict, prefix: str, ws: Tuple[int, int], heads: int, dtype=torch.float32) -> Tensor:
    """timm WindowAttention: relative_coords_table (offsets -(w-1)..(w-1) per axis / (w-1) * 8 -> sign * log2(|x| + 1) / log2 8) through
    cpb_mlp (Linear(2, 512), ReLU, Linear(512, heads, bias=False)), gathered by relative_position_index, 16 * sigmoid: [heads, N, N]."""
    dy = torch.arange(-(ws[0] - 1), ws[0], dtype=dtype) / max(ws[0] - 1, 1)
    dx = torch.arange(-(ws[1] - 1), ws[1], dtype=dtype) / max(ws[1] - 1, 1)
    tab = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), dim=-1) * 8.0
    tab = torch.sign(tab) * torch.log2(tab.abs() + 1.0) / math.log2(8.0)
    t = F.relu(F.linear(tab.reshape(-1, 2), sd[prefix + "cpb_mlp.0.weight"].to(dtype), sd[prefix + "cpb_mlp.0.bias"].to(dtype)))
    t = F.linear(t, sd[prefix + "cpb_mlp.2.weight"].to(dtype))                       # [(2wh-1)(2ww-1), heads]
    ys, xs = torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), indexing="ij")
    coords = torch.stack([ys, xs]).flatten(1)
    rel = coords[:, :, None] - coords[:, None, :]                                    # [2, N, N]
    idx = (rel[0] + ws[0] - 1) * (2 * ws[1] - 1) + (rel[1] + ws[1] - 1)
    n = ws[0] * ws[1]
    return 16.0 * torch.sigmoid(t[idx.reshape(-1)].reshape(n, n, heads).permute(2, 0, 1))


def shift_mask_timm(feat: Tuple[int, int], ws: Tuple[int, int], shift: Tuple[int, int], dtype=torch.float32) -> Optional[Tensor]:
    """timm SwinTransformerV2Block.__init__: img_mask labelled over the 3 x 3 slices (0:-ws, -ws:-shift, -shift:) of BOTH axes,
    window-partitioned; pairs with different labels get -100.  [num_windows, N, N] or None without a shift."""
    if not any(shift):
        return None
    H, W = feat
    img = torch.zeros(H, W, dtype=dtype)
    cnt = 0
    for hs in (slice(0, -ws[0]), slice(-ws[0], -shift[0]), slice(-shift[0], None)):
        for wsl in (slice(0, -ws[1]), slice(-ws[1], -shift[1]), slice(-shift[1], None)):
            img[hs, wsl] = cnt
            cnt += 1
    win = img.reshape(H // ws[0], ws[0], W // ws[1], ws[1]).permute(0, 2, 1, 3).reshape(-1, ws[0] * ws[1])
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def block_timm(x: Tensor, sd: dict, heads: int, ws: Tuple[int, int], shift: Tuple[int, int], prefix: str = "") -> Tensor:
    """timm SwinTransformerV2Block.forward on x [H, W, C] (B = 1): x + norm1(attn(x)); x + norm2(mlp(x)).  WindowAttention.forward:
    qkv = F.linear(x, qkv.weight, cat(q_bias, 0, v_bias)); cosine attention * exp(clamp(logit_scale, max = log 100)); + 16 sigmoid(cpb)
    bias; + mask; softmax; @ v; proj."""
    H, W, C = x.shape
    hd = C // heads
    n = ws[0] * ws[1]
    qkv_bias = torch.cat([sd[prefix + "attn.q_bias"], torch.zeros_like(sd[prefix + "attn.v_bias"]), sd[prefix + "attn.v_bias"]])
    qkv = F.linear(x, sd[prefix + "attn.qkv.weight"], qkv_bias)
    xs = torch.roll(qkv, shifts=(-shift[0], -shift[1]), dims=(0, 1)) if any(shift) else qkv
    win = xs.reshape(H // ws[0], ws[0], W // ws[1], ws[1], 3, heads, hd).permute(4, 0, 2, 5, 1, 3, 6).reshape(3, -1, heads, n, hd)
    q, k, v = win[0], win[1], win[2]
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    attn = attn * torch.clamp(sd[prefix + "attn.logit_scale"].reshape(1, heads, 1, 1), max=math.log(1.0 / 0.01)).exp()
    attn = attn + cpb_position_bias(sd, prefix + "attn.", ws, heads, x.dtype).unsqueeze(0)
    m = shift_mask_timm((H, W), ws, shift, x.dtype)
    if m is not None:
        attn = attn + m.unsqueeze(1)
    out = attn.softmax(dim=-1) @ v
    out = out.reshape(H // ws[0], W // ws[1], heads, ws[0], ws[1], hd).permute(0, 3, 1, 4, 2, 5).reshape(H, W, C)
    out = torch.roll(out, shifts=shift, dims=(0, 1)) if any(shift) else out
    a = F.linear(out, sd[prefix + "attn.proj.weight"], sd[prefix + "attn.proj.bias"])
    x = x + F.layer_norm(a, (C,), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], 1e-5)
    h = F.gelu(F.linear(x, sd[prefix + "mlp.fc1.weight"], sd[prefix + "mlp.fc1.bias"]))
    h = F.linear(h, sd[prefix + "mlp.fc2.weight"], sd[prefix + "mlp.fc2.bias"])
    return x + F.layer_norm(h, (C,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], 1e-5)


def stage_timm(x: Tensor, sd: dict, heads: int, window: int, depth: int, prefix: str = "blocks.") -> Tensor:
    """timm SwinTransformerV2Stage.forward without downsample: block i is shifted by window // 2 when i is odd; a window is clipped to
    the map and a clipped axis is not shifted (SwinTransformerV2Block._calc_window_shift)."""
    H, W, _ = x.shape
    ws = (min(window, H), min(window, W))
    for i in range(depth):
        shift = (0, 0) if i % 2 == 0 else tuple(0 if f <= s else window // 2 for f, s in zip((H, W), ws))
        x = block_timm(x, sd, heads, ws, shift, prefix=f"{prefix}{i}.")
    return x


def attend(q: Tensor, k: Tensor, v: Tensor, scale: Optional[float] = None) -> Tensor:
    """credit/attend.py:94-120 (Attend.forward, non-flash branch): softmax(q k^T * scale) v, q / k / v [b, h, n, d]."""
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (scale if scale is not None else q.shape[-1] ** -0.5)
    return torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
