"""CPU ORACLE (test infrastructure) for the input side of the step.

Restates credit/preblock/norm.py:78-98 (ERA5Normalizer._normalize_tensor: (t - mean) / std.clamp(min=1e-12), stats per
variable, scalar or one per level; variables without stats pass through) and credit/preblock/concat.py:22-31, 96-207
(ConcatToTensor: per source, variables stably sorted by (field-type rank, 3d before 2d), torch.cat along dim 1, channel map).
Pinned by tests/golden/preblock.npz (the reference classes run on a synthetic batch, tools/make_goldens.py --only pre).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch

FIELD_TYPE_RANK = {"prognostic": 0, "static": 1, "dynamic_forcing": 2, "diagnostic": 3}  # channel_utils.py:88-93


def sort_key(var_key: str):
    parts = var_key.split("/")
    ft = parts[1] if len(parts) > 1 else ""
    dim = parts[2] if len(parts) > 2 else ""
    return (FIELD_TYPE_RANK.get(ft, len(FIELD_TYPE_RANK)), 0 if dim == "3d" else 1)


def normalize(key: str, t: torch.Tensor, mean: Optional[Dict], std: Optional[Dict]) -> torch.Tensor:
    name = key.split("/")[-1]
    if mean is None or name not in mean:
        return t
    m = torch.as_tensor(mean[name], dtype=t.dtype)
    s = torch.as_tensor(std[name], dtype=t.dtype)
    if m.dim() == 1 and m.shape[0] > 1:
        m, s = m.view(1, -1, 1, 1, 1), s.view(1, -1, 1, 1, 1)
    return (t - m) / s.clamp(min=1e-12)


def assemble(batch_input: Dict[str, Dict], mean: Optional[Dict] = None, std: Optional[Dict] = None):
    tensors, cmap, cur = [], OrderedDict(), 0
    for _source, variables in batch_input.items():
        for k in sorted(variables.keys(), key=sort_key):
            t = normalize(k, variables[k], mean, std)
            tensors.append(t)
            nl, T = t.shape[1], t.shape[2]
            cmap[k] = {"slice": slice(cur, cur + nl * T), "orig_shape": (nl, T)}
            cur += nl * T
    return torch.cat(tensors, dim=1).float(), cmap
