"""CPU ORACLE — test infrastructure, not product code.

A functional restatement (torch CPU, fp32 or fp64) of ONE autoregressive forecast
step of CREDIT's legacy CrossFormer (`model.type: crossformer`), written from the
reference's algorithm, operating directly on a reference-layout state dict:

    earth pad -> 4 x (CrossEmbed -> Transformer) -> UpBlock x3 (+skip concat)
    -> ConvTranspose head -> unpad -> bilinear -> [tracer fixer]
    -> de-normalise -> next-input assembly

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the product path (`miles-credit_amd/wxengine`) never does and
fails loudly without its HIP library.

Pinning: `tools/make_goldens.py` runs the real reference (imported from
/root/reference in the dev container) on synthetic name-keyed weights and commits
its outputs under `tests/golden/`; `tests/test_oracle_golden.py` checks this
restatement against them, and `tests/test_oracle_vs_reference.py` checks it layer
by layer against the imported reference whenever /root/reference exists.

Each function cites the reference lines it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# weights
# --------------------------------------------------------------------------- #
def _as_t(a, dtype) -> Tensor:
    if isinstance(a, torch.Tensor):
        return a.detach().to(device="cpu", dtype=dtype)
    return torch.as_tensor(a).to(dtype)


def is_transposed_conv(prefix: str, sd: Optional[Dict] = None) -> bool:
    # credit/models/crossformer.py:92 (UpBlock.conv) and :572 (up_block4) are ConvTranspose2d; the wxformer
    # decoder (credit/models/wxformer/crossformer.py:137-162, 817-830) has only Conv2d (its state dict carries
    # "up_block1.sharp.*", which the legacy decoder never has)
    if sd is not None and any(k.startswith("up_block1.sharp.") for k in sd):
        return False
    # upsample_v_conv=True (credit/models/crossformer.py:87-89, 560-570): Upsample + Conv2d, up_block4 is a Sequential
    if sd is not None and any(k.startswith("up_block4.1.") for k in sd):
        return False
    return prefix == "up_block4" or (prefix.startswith("up_block") and prefix.endswith(".conv"))


def folded_weight(sd: Dict, prefix: str, dtype=torch.float32) -> Tensor:
    """Effective weight of a (possibly spectral-normed) Conv2d / Linear / ConvTranspose2d in eval mode.

    credit/models/crossformer.py:23-26 applies `nn.utils.spectral_norm`; in eval() its
    pre-forward hook computes W = weight_orig / sigma, sigma = u . (W_mat v), with W_mat the
    weight flattened after moving `dim` (0; 1 for ConvTranspose2d) to the front, and performs
    no power iteration (SURVEY.md Appendix A "Spectral norm").
    """
    if prefix + ".weight_orig" not in sd:
        return _as_t(sd[prefix + ".weight"], dtype)
    w = _as_t(sd[prefix + ".weight_orig"], dtype)
    u = _as_t(sd[prefix + ".weight_u"], dtype)
    v = _as_t(sd[prefix + ".weight_v"], dtype)
    w_mat = w.transpose(0, 1).reshape(w.shape[1], -1) if is_transposed_conv(prefix, sd) else w.reshape(w.shape[0], -1)
    sigma = torch.dot(u, torch.mv(w_mat, v))
    return w / sigma


def _bias(sd: Dict, prefix: str, dtype) -> Optional[Tensor]:
    k = prefix + ".bias"
    return _as_t(sd[k], dtype) if k in sd else None


# --------------------------------------------------------------------------- #
# a1: boundary padding
# --------------------------------------------------------------------------- #
def earth_pad_index(h: int, w: int, pad_lat, pad_lon):
    """Source (row, col) index maps of the earth padding, shape [Hp, Wp] each.

    credit/boundary_padding.py:50-72: roll lon by W//2, mirror the first p0 / last p1
    rows across the pole, then circular lon pad.  (SURVEY.md Appendix A.)
    """
    p0, p1 = pad_lat
    pl, pr = pad_lon
    hp, wp = h + p0 + p1, w + pl + pr
    i = torch.arange(hp).view(-1, 1).expand(hp, wp)
    j = torch.arange(wp).view(1, -1).expand(hp, wp)
    jj = (j - pl) % w
    rolled = (jj - w // 2) % w
    top = i < p0
    bot = i >= p0 + h
    src_r = torch.where(top, p0 - 1 - i, torch.where(bot, h - 1 - (i - p0 - h), i - p0))
    src_c = torch.where(top | bot, rolled, jj)
    return src_r, src_c


def earth_pad(x: Tensor, pad_lat, pad_lon) -> Tensor:
    """x [..., H, W] -> [..., H+p0+p1, W+pl+pr]."""
    if pad_lat[0] > 0 and pad_lat[1] == 0:
        raise ValueError("pad_lat=[p,0] hits the reference's `-0:` slicing quirk; not reproduced")
    src_r, src_c = earth_pad_index(x.shape[-2], x.shape[-1], pad_lat, pad_lon)
    return x[..., src_r, src_c]


def mirror_pad(x: Tensor, pad_lat, pad_lon) -> Tensor:
    """credit/boundary_padding.py:98-117 (mode "mirror"): circular pad in longitude, then reflect pad in latitude (the edge
    row is not repeated); no pole roll.  x [..., H, W] -> [..., H+p0+p1, W+pl+pr]."""
    h, w = x.shape[-2:]
    p0, p1 = pad_lat
    pl, pr = pad_lon
    i = torch.arange(h + p0 + p1)
    src_r = torch.where(i < p0, p0 - i, torch.where(i >= p0 + h, h - 2 - (i - p0 - h), i - p0))
    src_c = (torch.arange(w + pl + pr) - pl) % w
    return x[..., src_r.view(-1, 1), src_c.view(1, -1)]


def earth_unpad(x: Tensor, pad_lat, pad_lon) -> Tensor:
    # credit/boundary_padding.py:74-96
    h, w = x.shape[-2:]
    return x[..., pad_lat[0]: h - pad_lat[1], pad_lon[0]: w - pad_lon[1]]


# --------------------------------------------------------------------------- #
# a2: cross embed
# --------------------------------------------------------------------------- #
def cross_embed(x: Tensor, sd: Dict, prefix: str, kernels: List[int], stride: int, arch: str = "crossformer") -> Tensor:
    """credit/models/crossformer.py:128-152: one strided conv per (ascending) kernel size,
    padding (k-s)//2, outputs concatenated on channels.  arch "wxformer"
    (credit/models/wxformer/crossformer.py:199-236): explicit zero padding left/top (k-s)//2,
    right/bottom (k-s)-(k-s)//2, then an un-padded conv whose parameters sit at convs.<b>.1."""
    outs = []
    for b, k in enumerate(sorted(kernels)):
        if arch == "wxformer":
            p = f"{prefix}.convs.{b}.1"
            lo = (k - stride) // 2
            hi = (k - stride) - lo
            xp = F.pad(x, (lo, hi, lo, hi))
            outs.append(F.conv2d(xp, folded_weight(sd, p, x.dtype), _bias(sd, p, x.dtype), stride=stride))
        else:
            p = f"{prefix}.convs.{b}"
            outs.append(F.conv2d(x, folded_weight(sd, p, x.dtype), _bias(sd, p, x.dtype), stride=stride,
                                 padding=(k - stride) // 2))
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------- #
# a3: channel layer norm
# --------------------------------------------------------------------------- #
def channel_layernorm(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """credit/models/crossformer.py:182-192: biased variance over channels, eps inside sqrt."""
    mean = x.mean(dim=1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * g + b


# --------------------------------------------------------------------------- #
# a5: dynamic position bias
# --------------------------------------------------------------------------- #
def dpb_table(sd: Dict, prefix: str, wsz: int, dtype=torch.float32) -> Tensor:
    """MLP evaluated on the (2w+1)^2 integer offsets (credit/models/crossformer.py:158-176, :279-283)."""
    pos = torch.arange(-wsz, wsz + 1, dtype=dtype)
    rr, cc = torch.meshgrid(pos, pos, indexing="ij")
    t = torch.stack([rr.reshape(-1), cc.reshape(-1)], dim=1)  # [(2w+1)^2, 2]
    for lin, ln in ((0, 1), (3, 4), (6, 7)):
        p = f"{prefix}.layers.{lin}"
        t = F.linear(t, folded_weight(sd, p, dtype), _bias(sd, p, dtype))
        q = f"{prefix}.layers.{ln}"
        t = F.layer_norm(t, (t.shape[-1],), _as_t(sd[q + ".weight"], dtype), _as_t(sd[q + ".bias"], dtype), 1e-5)
        t = torch.relu(t)
    p = f"{prefix}.layers.9"
    t = F.linear(t, folded_weight(sd, p, dtype), _bias(sd, p, dtype))
    return t.reshape(-1)


def rel_pos_index(wsz: int) -> Tensor:
    """credit/models/crossformer.py:238-243: idx[i,j] = (dr+w-1)*(2w-1) + (dc+w-1)."""
    pos = torch.arange(wsz)
    rr, cc = torch.meshgrid(pos, pos, indexing="ij")
    r, c = rr.reshape(-1), cc.reshape(-1)
    dr = r[:, None] - r[None, :] + wsz - 1
    dc = c[:, None] - c[None, :] + wsz - 1
    return dr * (2 * wsz - 1) + dc


def dpb_bias(sd: Dict, prefix: str, wsz: int, dtype=torch.float32) -> Tensor:
    """[N, N] bias, N = wsz^2.  Reproduces the reference's indexing of a stride-(2w+1) table
    with stride-(2w-1) indices (credit/models/crossformer.py:284)."""
    return dpb_table(sd, prefix, wsz, dtype)[rel_pos_index(wsz)]


# --------------------------------------------------------------------------- #
# a4: window attention
# --------------------------------------------------------------------------- #
def window_partition(x: Tensor, wsz: int, kind: str) -> Tensor:
    """x [C,H,W] -> tokens [nWin, N, C]; window order row-major over (h, w), token order row-major.

    short: contiguous wsz x wsz blocks; long: dilated grid with stride (H/wsz, W/wsz)
    (credit/models/crossformer.py:261-264)."""
    c, h, w = x.shape
    if kind == "short":
        t = x.reshape(c, h // wsz, wsz, w // wsz, wsz).permute(1, 3, 2, 4, 0)
    else:
        t = x.reshape(c, wsz, h // wsz, wsz, w // wsz).permute(2, 4, 1, 3, 0)
    return t.reshape(-1, wsz * wsz, c)


def window_merge(t: Tensor, h: int, w: int, wsz: int, kind: str) -> Tensor:
    """inverse of window_partition -> [C,H,W] (credit/models/crossformer.py:301-314)."""
    c = t.shape[-1]
    if kind == "short":
        return t.reshape(h // wsz, w // wsz, wsz, wsz, c).permute(4, 0, 2, 1, 3).reshape(c, h, w)
    return t.reshape(h // wsz, w // wsz, wsz, wsz, c).permute(4, 2, 0, 3, 1).reshape(c, h, w)


def attention(x: Tensor, sd: Dict, prefix: str, kind: str, wsz: int, dim_head: int = 32) -> Tensor:
    """Attention.forward without the residual (credit/models/crossformer.py:247-316). x [1,C,H,W]."""
    _, c, h, w = x.shape
    heads = c // dim_head
    xn = channel_layernorm(x, _as_t(sd[prefix + ".norm.g"], x.dtype), _as_t(sd[prefix + ".norm.b"], x.dtype))
    tok = window_partition(xn[0], wsz, kind)  # [nW, N, C]
    wqkv = folded_weight(sd, prefix + ".to_qkv", x.dtype).reshape(3 * c, c)
    qkv = tok @ wqkv.t()  # [nW, N, 3C]
    nw, n, _ = qkv.shape
    q, k, v = (qkv[..., i * c:(i + 1) * c].reshape(nw, n, heads, dim_head).permute(0, 2, 1, 3) for i in range(3))
    q = q * (dim_head ** -0.5)
    sim = q @ k.transpose(-1, -2) + dpb_bias(sd, prefix + ".dpb", wsz, x.dtype)
    attn = torch.softmax(sim, dim=-1)
    out = (attn @ v).permute(0, 2, 1, 3).reshape(nw, n, c)
    wout = folded_weight(sd, prefix + ".to_out", x.dtype).reshape(c, c)
    out = out @ wout.t() + _bias(sd, prefix + ".to_out", x.dtype)
    return window_merge(out, h, w, wsz, kind).unsqueeze(0)


# --------------------------------------------------------------------------- #
# a6: feed forward
# --------------------------------------------------------------------------- #
def feedforward(x: Tensor, sd: Dict, prefix: str) -> Tensor:
    """FeedForward without the residual (credit/models/crossformer.py:195-207): LN, 1x1 C->4C, exact GELU, 1x1 4C->C."""
    xn = channel_layernorm(x, _as_t(sd[prefix + ".layers.0.g"], x.dtype), _as_t(sd[prefix + ".layers.0.b"], x.dtype))
    h1 = F.conv2d(xn, folded_weight(sd, prefix + ".layers.1", x.dtype), _bias(sd, prefix + ".layers.1", x.dtype))
    h1 = 0.5 * h1 * (1.0 + torch.erf(h1 / math.sqrt(2.0)))
    return F.conv2d(h1, folded_weight(sd, prefix + ".layers.4", x.dtype), _bias(sd, prefix + ".layers.4", x.dtype))


def transformer(x: Tensor, sd: Dict, prefix: str, depth: int, local_w: int, global_w: int, dim_head: int,
                capture=None) -> Tensor:
    """credit/models/crossformer.py:358-365."""
    for d in range(depth):
        p = f"{prefix}.layers.{d}"
        x = attention(x, sd, p + ".0", "short", local_w, dim_head) + x
        if capture is not None:
            capture[p + ".0"] = x
        x = feedforward(x, sd, p + ".1") + x
        if capture is not None:
            capture[p + ".1"] = x
        x = attention(x, sd, p + ".2", "long", global_w, dim_head) + x
        if capture is not None:
            capture[p + ".2"] = x
        x = feedforward(x, sd, p + ".3") + x
        if capture is not None:
            capture[p + ".3"] = x
    return x


# --------------------------------------------------------------------------- #
# a8: decoder
# --------------------------------------------------------------------------- #
def group_norm_silu(x: Tensor, weight: Tensor, bias: Tensor, groups: int, eps: float = 1e-5) -> Tensor:
    """nn.GroupNorm(groups, C) then SiLU (credit/models/crossformer.py:98-100): biased variance over (C/G,H,W)."""
    b, c, h, w = x.shape
    xg = x.reshape(b, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    xn = ((xg - mean) / torch.sqrt(var + eps)).reshape(b, c, h, w)
    y = xn * weight.view(1, c, 1, 1) + bias.view(1, c, 1, 1)
    return y * torch.sigmoid(y)


def upsample2x(x: Tensor) -> Tensor:
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False) restated (credit/models/crossformer.py:88):
    src = max((dst + 0.5) / 2 - 0.5, 0) -> weights 0.75 / 0.25 with the border rows / columns replicated."""
    def axis(t: Tensor, dim: int) -> Tensor:
        n = t.shape[dim]
        idx = torch.arange(n)
        lo = t.index_select(dim, (idx - 1).clamp(min=0))
        hi = t.index_select(dim, (idx + 1).clamp(max=n - 1))
        even = 0.25 * lo + 0.75 * t      # dst 2i   : src = i - 0.25
        odd = 0.75 * t + 0.25 * hi       # dst 2i+1 : src = i + 0.25
        out = torch.stack([even, odd], dim=dim + 1)
        shape = list(t.shape)
        shape[dim] = 2 * n
        return out.reshape(shape)
    return axis(axis(x, x.dim() - 2), x.dim() - 1)


def up_block(x: Tensor, sd: Dict, prefix: str, groups: int, upconv: bool = False) -> Tensor:
    """UpBlock.forward (credit/models/crossformer.py:107-122), attention None; upconv = upsample_v_conv."""
    if upconv:
        x = F.conv2d(upsample2x(x), folded_weight(sd, prefix + ".conv", x.dtype), _bias(sd, prefix + ".conv", x.dtype), padding=1)
    else:
        x = F.conv_transpose2d(x, folded_weight(sd, prefix + ".conv", x.dtype), _bias(sd, prefix + ".conv", x.dtype),
                               stride=2)
    shortcut = x
    for j in (0, 3):
        p = f"{prefix}.b.{j}"
        x = F.conv2d(x, folded_weight(sd, p, x.dtype), _bias(sd, p, x.dtype), padding=1)
        q = f"{prefix}.b.{j + 1}"
        x = group_norm_silu(x, _as_t(sd[q + ".weight"], x.dtype), _as_t(sd[q + ".bias"], x.dtype), groups)
    return x + shortcut


def pixel_shuffle2(x: Tensor) -> Tensor:
    """nn.PixelShuffle(2): channel c*4 + 2*dy + dx -> pixel (2y+dy, 2x+dx) of channel c."""
    b, c4, h, w = x.shape
    c = c4 // 4
    return x.reshape(b, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(b, c, 2 * h, 2 * w)


def up_block_ps(x: Tensor, sd: Dict, prefix: str, groups: int) -> Tensor:
    """UpBlockPS.forward (credit/models/wxformer/crossformer.py:156-162): sub-pixel conv + PixelShuffle,
    x += sharp(x), then the same residual stack as the legacy block."""
    x = pixel_shuffle2(F.conv2d(x, folded_weight(sd, prefix + ".conv", x.dtype), _bias(sd, prefix + ".conv", x.dtype),
                                padding=1))
    x = x + F.conv2d(x, folded_weight(sd, prefix + ".sharp", x.dtype), _bias(sd, prefix + ".sharp", x.dtype), padding=1)
    shortcut = x
    for j in (0, 3):
        p = f"{prefix}.b.{j}"
        x = F.conv2d(x, folded_weight(sd, p, x.dtype), _bias(sd, p, x.dtype), padding=1)
        q = f"{prefix}.b.{j + 1}"
        x = group_norm_silu(x, _as_t(sd[q + ".weight"], x.dtype), _as_t(sd[q + ".bias"], x.dtype), groups)
    return x + shortcut


# --------------------------------------------------------------------------- #
# a10: head
# --------------------------------------------------------------------------- #
def bilinear_resize(x: Tensor, out_h: int, out_w: int) -> Tensor:
    """F.interpolate(mode='bilinear', align_corners=False) restated (credit/models/crossformer.py:632;
    SURVEY.md Appendix A): src = max((dst+0.5)*in/out - 0.5, 0)."""
    def axis(n_in, n_out):
        # ATen (UpSample.h area_pixel_compute_source_index) does this index math in fp32 and its
        # CPU build contracts scale*(d+0.5)-0.5 into ONE fma (measured here: 4.8e-7 vs 2.6e-5 max
        # deviation from F.interpolate at 720->721), so round once: fp64 product, then fp32.
        d = torch.arange(n_out, dtype=torch.float32)
        scale = torch.tensor(n_in, dtype=torch.float32) / torch.tensor(n_out, dtype=torch.float32)
        src = (scale.double() * (d + 0.5).double() - 0.5).float().clamp(min=0.0)
        i0 = torch.floor(src).to(torch.long).clamp(max=n_in - 1)
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        lam = (src - i0.to(torch.float32)).to(x.dtype)
        return i0, i1, lam
    r0, r1, lr = axis(x.shape[-2], out_h)
    c0, c1, lc = axis(x.shape[-1], out_w)
    rows = x[..., r0, :] * (1 - lr).view(-1, 1) + x[..., r1, :] * lr.view(-1, 1)
    return rows[..., c0] * (1 - lc) + rows[..., c1] * lc


# --------------------------------------------------------------------------- #
# forward
# --------------------------------------------------------------------------- #
def forward(cfg, sd: Dict, x, dtype=torch.float32, capture: Optional[Dict] = None) -> Tensor:
    """CrossFormer.forward (credit/models/crossformer.py:593-644) without the in-model PostBlock.

    x: [B, C_in, frames, H, W] (or [B, C, H, W] when frames == 1) -> [B, C_out, out_frames, H, W].
    """
    x = _as_t(x, dtype)
    if x.dim() == 4:
        x = x.unsqueeze(2)
    if cfg.pad_activate:
        x = (mirror_pad if getattr(cfg, "pad_mode", "earth") == "mirror" else earth_pad)(x, cfg.pad_lat, cfg.pad_lon)
    b, c, t, h, w = x.shape
    x = x.reshape(b, c * t, h, w)  # frames==1: squeeze(2); frames>1: channel-major then time (:604-609)
    if capture is not None:
        capture["pad"] = x
    outs = []
    for bi in range(b):
        z = x[bi:bi + 1]
        enc = []
        for s in range(4):
            z = cross_embed(z, sd, f"layers.{s}.0", list(cfg.cross_embed_kernel_sizes[s]), cfg.cross_embed_strides[s],
                            getattr(cfg, "arch", "crossformer"))
            if capture is not None and bi == 0:
                capture[f"layers.{s}.0"] = z
            z = transformer(z, sd, f"layers.{s}.1", cfg.depth[s], cfg.local_window_size[s],
                            cfg.global_window_size[s], cfg.dim_head, capture if bi == 0 else None)
            if capture is not None and bi == 0:
                capture[f"layers.{s}.1"] = z
            enc.append(z)
        g = cfg.dim[0]
        upconv = bool(getattr(cfg, "upsample_v_conv", False))
        if getattr(cfg, "arch", "crossformer") == "wxformer":
            ub = up_block_ps
        else:
            def ub(t, sd_, prefix, groups):
                return up_block(t, sd_, prefix, groups, upconv)
        z = ub(z, sd, "up_block1", g)
        if capture is not None and bi == 0:
            capture["up_block1"] = z
        z = ub(torch.cat([z, enc[2]], dim=1), sd, "up_block2", g)
        if capture is not None and bi == 0:
            capture["up_block2"] = z
        z = ub(torch.cat([z, enc[1]], dim=1), sd, "up_block3", g)
        if capture is not None and bi == 0:
            capture["up_block3"] = z
        z = torch.cat([z, enc[0]], dim=1)
        if getattr(cfg, "arch", "crossformer") == "wxformer":  # wxformer/crossformer.py:817-830
            z = pixel_shuffle2(F.conv2d(z, folded_weight(sd, "up_block4.0", dtype), _bias(sd, "up_block4.0", dtype), padding=1))
            z = F.conv2d(z, folded_weight(sd, "up_block4.2", dtype), _bias(sd, "up_block4.2", dtype), padding=1)
        elif upconv:  # crossformer.py:560-570
            z = F.conv2d(upsample2x(z), folded_weight(sd, "up_block4.1", dtype), _bias(sd, "up_block4.1", dtype), padding=1)
        else:
            z = F.conv_transpose2d(z, folded_weight(sd, "up_block4", dtype), _bias(sd, "up_block4", dtype), stride=2,
                                   padding=1)
        if capture is not None and bi == 0:
            capture["up_block4"] = z
        if cfg.pad_activate:
            z = earth_unpad(z, cfg.pad_lat, cfg.pad_lon)
        if cfg.interp:
            z = bilinear_resize(z, cfg.image_height, cfg.image_width)
        outs.append(z)
    y = torch.cat(outs, dim=0)
    return y.reshape(b, cfg.base_output_channels, cfg.output_frames, y.shape[-2], y.shape[-1])


# --------------------------------------------------------------------------- #
# a12-a14: step glue
# --------------------------------------------------------------------------- #
def tracer_fix(y: Tensor, tracer_inds, thres, mean: Optional[Tensor] = None, std: Optional[Tensor] = None,
               thres_max=None) -> Tensor:
    """TracerFixer.forward (credit/postblock/gen1.py:136-167): optional de-normalise, clamp
    y[:, i] < thres -> thres (and >= max -> max), re-normalise.  mean/std are per-output-channel;
    None means `denorm: False`."""
    y = y.clone()
    if mean is not None:
        y = y * std.view(1, -1, 1, 1, 1) + mean.view(1, -1, 1, 1, 1)
    for n, i in enumerate(tracer_inds):
        v = y[:, i]
        v[v < thres[n]] = thres[n]
        if thres_max is not None:
            v[v >= thres_max[n]] = thres_max[n]
    if mean is not None:
        y = (y - mean.view(1, -1, 1, 1, 1)) / std.view(1, -1, 1, 1, 1)
    return y


def denorm(y: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """credit/applications/rollout_to_netcdf.py:287: y*std+mean, time axis squeezed."""
    return (y * std.view(1, -1, 1, 1, 1) + mean.view(1, -1, 1, 1, 1)).squeeze(2)


def update_x(x_prev: Tensor, x_frc: Tensor, y: Tensor, n_prog: int, n_static: int) -> Tensor:
    """Single-source next-input assembly (credit/datasets/gen_2/channel_utils.py:253-291):
    x = [prognostic | static | dynamic_forcing]; prognostic <- y[:, :n_prog], forcing <- x_frc."""
    x = x_prev.clone()
    x[:, :n_prog] = y[:, :n_prog]
    n_dyn = x_frc.shape[1]
    x[:, n_prog + n_static: n_prog + n_static + n_dyn] = x_frc
    return x


def update_x_groups(x_prev: Tensor, x_frc: Tensor, y: Tensor, groups) -> Tensor:
    """Next-input assembly for ANY number of data sources (credit/datasets/gen_2/channel_utils.py:253-291 with the ChannelGroup
    list of :140-250).  groups: (field_type | code, x_start, src_start or None / -1, count); prognostic (0) <- y, dynamic_forcing
    (1) <- x_frc, anything else is carried forward."""
    x = x_prev.clone()
    for kind, x0, s0, n in groups:
        if kind in ("prognostic", 0):
            x[:, x0:x0 + n] = y[:, s0:s0 + n]
        elif kind in ("dynamic_forcing", 1):
            x[:, x0:x0 + n] = x_frc[:, s0:s0 + n]
    return x


def rollout(cfg, sd, x0, forcings, n_static: int, mean=None, std=None, tracer=None, dtype=torch.float32):
    """predict()'s hot loop (credit/applications/rollout_to_netcdf.py:274-310) on synthetic forcing.

    Returns (list of y_t normalised, list of y_phys_t).  `tracer` = dict(inds, thres, denorm) or None.
    """
    x = _as_t(x0, dtype)
    n_prog = cfg.channels * cfg.levels + cfg.surface_channels
    mean_t = _as_t(mean, dtype) if mean is not None else torch.zeros(cfg.base_output_channels, dtype=dtype)
    std_t = _as_t(std, dtype) if std is not None else torch.ones(cfg.base_output_channels, dtype=dtype)
    ys, phys = [], []
    for t, frc in enumerate(forcings):
        y = forward(cfg, sd, x, dtype)
        if tracer is not None:
            use = tracer.get("denorm", True)
            y = tracer_fix(y, tracer["inds"], tracer["thres"], mean_t if use else None, std_t if use else None)
        ys.append(y)
        phys.append(denorm(y, mean_t, std_t))
        if frc is not None:
            x = update_x(x, _as_t(frc, dtype), y, n_prog, n_static)
    return ys, phys
