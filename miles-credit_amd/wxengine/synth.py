"""Synthetic, name-keyed weights and inputs (SURVEY.md §8(c) "Fixture plan").

Real checkpoints cannot be shipped (25-124 M parameters) and torch's init order
is not reproducible without the reference code, so every tensor is generated
from a counter-based RNG (numpy Philox) keyed by the FNV-1a hash of its
state-dict key.  The same generator feeds the reference (when building golden
fixtures), the oracle and the engine, so all three see bit-identical weights.

Spectral-norm caveat (SURVEY.md header): `eval()` uses the stored `weight_u`,
`weight_v` without power iteration, so random u/v make activations explode.  We
store u/v after `POWER_ITERS` power iterations on `weight_orig`, which is what a
trained checkpoint holds.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

POWER_ITERS = 6
_FNV_OFFSET = 0xCBF29CE484222325
_FNV_PRIME = 0x100000001B3


def fnv1a64(text: str) -> int:
    h = _FNV_OFFSET
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * _FNV_PRIME) & 0xFFFFFFFFFFFFFFFF
    return h


def keyed_normal(key: str, shape, seed: int = 0) -> np.ndarray:
    """float32 N(0,1) tensor that depends only on (key, seed, shape)."""
    g = np.random.Generator(np.random.Philox(key=[fnv1a64(key), seed & 0xFFFFFFFFFFFFFFFF]))
    return g.standard_normal(size=shape, dtype=np.float32)


def is_transposed_conv(prefix: str, arch: str = "crossformer", upconv: bool = False) -> bool:
    """ConvTranspose2d modules of the legacy decoder (reference crossformer.py:92,572):
    their spectral norm is taken over weight dim 1.  The wxformer decoder and the upsample_v_conv variant have none."""
    if arch == "wxformer" or upconv:
        return False
    return prefix == "up_block4" or (prefix.startswith("up_block") and prefix.endswith(".conv"))


def _l2n(v, eps=1e-12):
    return v / max(float(np.linalg.norm(v)), eps)


def power_iterate(w_mat: np.ndarray, u0: np.ndarray, iters: int = POWER_ITERS):
    """torch.nn.utils.spectral_norm's power iteration (fp32), returns (u, v)."""
    u = _l2n(u0.astype(np.float32))
    v = None
    for _ in range(iters):
        v = _l2n(w_mat.T @ u)
        u = _l2n(w_mat @ v)
    return u.astype(np.float32), v.astype(np.float32)


def synth_state_dict(cfg, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Full reference-layout state dict (numpy float32) for `cfg.state_spec()`."""
    spec = cfg.state_spec()
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in spec.items():
        if key.endswith((".weight_u", ".weight_v")):
            continue  # filled below with their weight_orig
        z = keyed_normal(key, shape, seed)
        if key.endswith(".weight_orig") or (key.endswith(".weight") and len(shape) >= 2):
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            sd[key] = (z / np.sqrt(max(fan_in, 1))).astype(np.float32)
        elif key.endswith((".g", ".weight")) and (len(shape) == 1 or key.endswith(".g")):
            sd[key] = (1.0 + 0.1 * z).astype(np.float32)  # LN/GN gains
        else:
            sd[key] = (0.1 * z).astype(np.float32)  # biases, LN/GN shifts
    for key, shape in spec.items():
        if not key.endswith(".weight_u"):
            continue
        base = key[: -len(".weight_u")]
        w = sd[base + ".weight_orig"]
        transposed = is_transposed_conv(base, getattr(cfg, "arch", "crossformer"), bool(getattr(cfg, "upsample_v_conv", False)))
        if transposed:  # ConvTranspose2d: spectral_norm(dim=1)
            w_mat = np.moveaxis(w, 1, 0).reshape(w.shape[1], -1)
        else:
            w_mat = w.reshape(w.shape[0], -1)
        u, v = power_iterate(w_mat, keyed_normal(key, (w_mat.shape[0],), seed))
        sd[base + ".weight_u"] = u
        sd[base + ".weight_v"] = v
    return OrderedDict((k, sd[k]) for k in spec)


def synth_input(cfg, seed: int = 1000) -> np.ndarray:
    """x0 ~ N(0,1), float32 [1, C_in, frames, H, W] (SURVEY.md §8(d) 'Synthetic inputs')."""
    return keyed_normal("x0", (1, cfg.base_input_channels, cfg.frames, cfg.image_height, cfg.image_width), seed)


def synth_forcing(cfg, n_dyn: int, step: int, seed: int = 1000) -> np.ndarray:
    """Per-step dynamic forcing, float32 [1, n_dyn, 1, H, W]."""
    return keyed_normal(f"frc{step}", (1, n_dyn, 1, cfg.image_height, cfg.image_width), seed)


def synth_denorm(n_out: int, seed: int = 0):
    """Per-output-channel (mean, std) with std in [0.5, 2], mean in [-1, 1]."""
    z = keyed_normal("denorm", (2, n_out), seed)
    mean = np.tanh(z[0]).astype(np.float32)
    std = (0.5 + 1.5 / (1.0 + np.exp(-z[1]))).astype(np.float32)
    return mean, std
