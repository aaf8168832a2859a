"""Input side of the step on the device: host mirror of the reference's preblocks.

`DevicePreblock` takes the nested batch dict the reference's dataloader produces
(`batch["input"][source][var_key] -> tensor [B, n_levels, T, H, W]`, var_key = "source/field_type/dim/varname"),
orders the variables exactly like `credit/preblock/concat.py::_channel_sort_key` (:22-31: field-type rank
prognostic < static < dynamic_forcing < diagnostic from `credit/datasets/gen_2/channel_utils.py:88-93`, 3d before 2d,
otherwise insertion order -- Python's sort is stable), normalises them like
`credit/preblock/norm.py::ERA5Normalizer._normalize_tensor` (:78-98) and concatenates along the channel dim like
`ConcatToTensor.forward` (:96-207) -- the last two in ONE kernel pass through the C ABI (`wx_pre_*`).  It also returns the
reference's `_channel_map` (var_key -> {"slice", "orig_shape"}).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np

from .engine import WXEngineError, _check, load_library

FIELD_TYPE_RANK = {"prognostic": 0, "static": 1, "dynamic_forcing": 2, "diagnostic": 3}


def channel_sort_key(var_key: str):
    parts = var_key.split("/")
    ft = parts[1] if len(parts) > 1 else ""
    dim = parts[2] if len(parts) > 2 else ""
    return (FIELD_TYPE_RANK.get(ft, len(FIELD_TYPE_RANK)), 0 if dim == "3d" else 1)


def ordered_keys(sources: Dict[str, Dict]) -> list:
    """Concatenation order of `batch["input"]`: sources in insertion order, variables sorted (stably) inside each."""
    out = []
    for _source, variables in sources.items():
        out += sorted(variables.keys(), key=channel_sort_key)
    return out


def channel_stats(keys, levels, mean: Optional[Dict], std: Optional[Dict]):
    """Per-output-channel (mean, std) from per-variable stats (scalar or per-level vectors); variables without stats
    pass through unchanged (norm.py:84-85) = mean 0, std 1."""
    if mean is None:
        return None, None
    m, s = [], []
    for k, nl in zip(keys, levels):
        name = k.split("/")[-1]
        if name in mean:
            mv, sv = np.asarray(mean[name], np.float32).ravel(), np.asarray(std[name], np.float32).ravel()
            if mv.size == 1:
                mv, sv = np.repeat(mv, nl), np.repeat(sv, nl)
            if mv.size != nl:
                raise ValueError(f"{k}: {mv.size} statistics for {nl} levels")
        else:
            mv, sv = np.zeros(nl, np.float32), np.ones(nl, np.float32)
        m.append(mv)
        s.append(sv)
    return np.concatenate(m).astype(np.float32), np.concatenate(s).astype(np.float32)


class DevicePreblock:
    def __init__(self, example_input: Dict[str, Dict], mean: Optional[Dict] = None, std: Optional[Dict] = None,
                 device: Optional[int] = None):
        """`device`: GPU index the block lives on; default = the device of the example tensors when they are on a GPU, else the
        current device (rank r of a replicas run works on cuda:r -- a block pinned to GPU 0 would launch on another device's
        stream there)."""
        import torch
        if not torch.cuda.is_available():
            raise WXEngineError("no GPU visible: the device preblock has no CPU fallback")
        self.lib = load_library()
        self.keys = ordered_keys(example_input)
        flat = {k: v for src in example_input.values() for k, v in src.items()}
        if device is None:
            on_gpu = [v.device.index for v in flat.values() if getattr(v, "is_cuda", False)]
            device = on_gpu[0] if on_gpu else torch.cuda.current_device()
        self.device = int(device)
        shp = [tuple(flat[k].shape) for k in self.keys]
        if any(len(s) != 5 for s in shp) or len({(s[2], s[3], s[4]) for s in shp}) != 1:
            raise ValueError("fields must be [B, n_levels, T, H, W] on one grid")
        self.levels = [s[1] for s in shp]
        self.T, self.H, self.W = shp[0][2:]
        self.mean, self.std = channel_stats(self.keys, self.levels, mean, std)
        self.channel_map = OrderedDict()
        cur = 0
        for k, nl in zip(self.keys, self.levels):
            # ConcatToTensor flattens (levels x T) per variable in its map; the tensor itself stays [B, C, T, H, W]
            self.channel_map[k] = {"slice": slice(cur, cur + nl * self.T), "orig_shape": (nl, self.T)}
            cur += nl * self.T
        self.channels = sum(self.levels)
        lv = (C.c_int32 * len(self.levels))(*self.levels)
        fp = C.POINTER(C.c_float)
        self._p = C.c_void_p()
        _check(self.lib.wx_pre_create(len(self.levels), lv, self.T, self.H, self.W,
                                      self.mean.ctypes.data_as(fp) if self.mean is not None else None,
                                      self.std.ctypes.data_as(fp) if self.std is not None else None, self.device, C.byref(self._p)))

    def __del__(self):
        try:
            if getattr(self, "_p", None):
                self.lib.wx_pre_destroy(self._p)
                self._p = C.c_void_p()
        except Exception:
            pass

    def __call__(self, batch_input: Dict[str, Dict]):
        """-> x [B, C, T, H, W] float32 on the GPU (normalised + concatenated)."""
        import torch
        flat = {k: v for src in batch_input.values() for k, v in src.items()}
        ts = []
        for k, nl in zip(self.keys, self.levels):
            t = flat[k]
            if not t.is_cuda:
                t = t.cuda(self.device, non_blocking=True)
            elif t.device.index != self.device:
                raise WXEngineError(f"{k} is on cuda:{t.device.index} but this preblock was created for cuda:{self.device}")
            t = t.contiguous().float()
            if tuple(t.shape[1:]) != (nl, self.T, self.H, self.W):
                raise ValueError(f"{k}: shape {tuple(t.shape)} does not match the schema")
            ts.append(t)
        B = ts[0].shape[0]
        x = torch.empty((B, self.channels, self.T, self.H, self.W), dtype=torch.float32, device=ts[0].device)
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        with torch.cuda.device(self.device):   # the entry point selects its own device; keep torch's notion of "current" intact
            _check(self.lib.wx_pre_apply(self._p, ptrs, C.c_void_p(x.data_ptr()), B,
                                         C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return x
