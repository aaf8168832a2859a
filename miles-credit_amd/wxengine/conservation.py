"""Gen-2 (name-keyed) conservation post-blocks on the device -- SURVEY.md §8(f) row 1, the conservation part.

Same constructor arguments, same `forward(batch_dict) -> batch_dict` contract and the same variable addressing as
`credit/postblock/conservation.py` (TracerFixer :84-114, GlobalMassFixer :117-176, GlobalWaterFixer :179-236,
GlobalEnergyFixerUpDown :239-376): predictions are read from `batch_dict["y_processed"][source][var_key]`
([B, L, 1, H, W], physical units), the t0 state from `batch_dict[input_source_key][source][var_key]` (last frame), and the
corrected variable is written back under the same key.  The arithmetic runs in the engine's device PostBlock
(`wx_post_*`): the named tensors are stacked into one channel block, fixed in place, and the touched variable is sliced
back out.  The physics arrays the reference reads from `save_loc_physics` (an xarray file) are passed in directly:
`physics = dict(lat2d=..., lon2d=..., coef_a=..., coef_b=..., midpoint=..., [gph_surf=...])`.  No CPU fallback.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .engine import WXPostBlock


def _src(var_key: str) -> str:
    return var_key.split("/")[0]


def _pred(batch_dict, var_key):
    return batch_dict["y_processed"][_src(var_key)][var_key]


def _set_pred(batch_dict, var_key, t):
    batch_dict["y_processed"][_src(var_key)][var_key] = t


class TracerFixer:
    """credit/postblock/conservation.py:84-114 (a clamp; torch on the device tensor is already the fused form)."""

    def __init__(self, tracer_vars, tracer_thres, tracer_thres_max=None):
        self.tracer_vars = list(tracer_vars)
        n = len(self.tracer_vars)
        self.lo = list(tracer_thres) if isinstance(tracer_thres, (list, tuple)) else [tracer_thres] * n
        if tracer_thres_max is None:
            self.hi = [None] * n
        else:
            self.hi = list(tracer_thres_max) if isinstance(tracer_thres_max, (list, tuple)) else [tracer_thres_max] * n

    def __call__(self, batch_dict):
        for k, lo, hi in zip(self.tracer_vars, self.lo, self.hi):
            t = torch.clamp(_pred(batch_dict, k), min=lo)
            if hi is not None:
                t = torch.clamp(t, max=hi)
            _set_pred(batch_dict, k, t)
        return batch_dict

    forward = __call__


class _DeviceFixer:
    """Stacks named variables into the channel blocks the device PostBlock works on and slices the result back."""

    def __init__(self, input_source_key: str, physics: Dict):
        self.input_source_key = input_source_key
        self.ph = physics
        self._pb: Optional[WXPostBlock] = None

    def _input(self, batch_dict, k):
        return batch_dict[self.input_source_key][_src(k)][k]

    def _stack(self, batch_dict, pred_keys: List[str], input_keys: List[str]):
        """y [B, C, H, W] from the predictions (+ input-only extras appended by the caller), x [B, C, 1, H, W] last frame."""
        ys = [_pred(batch_dict, k)[:, :, 0] for k in pred_keys]
        dev = ys[0].device
        xs = [self._input(batch_dict, k)[:, :, -1].to(dev) for k in input_keys]
        return ys, xs

    def _post(self, c_in: int, c_out: int, H: int, W: int, sp_ind: int, dev_index: int) -> WXPostBlock:
        if self._pb is None:
            pb = WXPostBlock(H, W, c_in, 1, c_out, dev_index)
            ph = self.ph
            if ph.get("grid_type", "sigma") == "sigma":
                pb.set_grid_sigma(ph["lat2d"], ph["lon2d"], ph["coef_a"], ph["coef_b"], sp_ind, bool(ph.get("midpoint", False)))
            else:
                pb.set_grid(ph["lat2d"], ph["lon2d"], ph["p_levels"], bool(ph.get("midpoint", False)))
            self._build(pb)
            self._pb = pb
        return self._pb

    def _run(self, y: torch.Tensor, x: torch.Tensor, sp_ind: int) -> torch.Tensor:
        y = y.contiguous().float()
        x = x.contiguous().float()
        pb = self._post(x.shape[1], y.shape[1], y.shape[-2], y.shape[-1], sp_ind, y.device.index or 0)
        for b in range(y.shape[0]):
            pb.apply(x[b].unsqueeze(1), y[b])
        return y


class GlobalMassFixer(_DeviceFixer):
    def __init__(self, q_var, sp_var, input_source_key="x_physical", **physics):
        super().__init__(input_source_key, physics)
        self.q_var, self.sp_var = q_var, sp_var

    def _build(self, pb):
        pb.add_mass_fixer(0, 1)

    def __call__(self, batch_dict):
        ys, xs = self._stack(batch_dict, [self.q_var, self.sp_var], [self.q_var, self.sp_var])
        L = ys[0].shape[1]
        y = self._run(torch.cat(ys, 1), torch.cat(xs, 1), L)
        _set_pred(batch_dict, self.sp_var, y[:, L:L + 1].unsqueeze(2))
        return batch_dict

    forward = __call__


class GlobalWaterFixer(_DeviceFixer):
    def __init__(self, q_var, sp_var, precip_var, evapor_var, lead_time_periods, input_source_key="x_physical", **physics):
        super().__init__(input_source_key, physics)
        self.q_var, self.sp_var, self.precip_var, self.evapor_var = q_var, sp_var, precip_var, evapor_var
        self.n_seconds = int(lead_time_periods) * 3600

    def _build(self, pb):
        L = self._L
        pb.add_water_fixer(0, L + 1, L + 2, self.n_seconds)

    def __call__(self, batch_dict):
        ys, xs = self._stack(batch_dict, [self.q_var, self.sp_var, self.precip_var, self.evapor_var], [self.q_var, self.sp_var])
        self._L = L = ys[0].shape[1]
        y = self._run(torch.cat(ys, 1), torch.cat(xs, 1), L)
        _set_pred(batch_dict, self.precip_var, y[:, L + 1:L + 2].unsqueeze(2))
        return batch_dict

    forward = __call__


class GlobalEnergyFixerUpDown(_DeviceFixer):
    """conservation.py:239-376.  R_T = (SOLIN*dt - USW*dt - OLR*dt)/dt with SOLIN read from the INPUT dict; F_S =
    (FSDS - FSUS + FLDS - FLUS + SHF + LHF)/dt (note the + on the turbulent fluxes, unlike gen1's up/down class)."""

    def __init__(self, T_var, q_var, U_var, V_var, sp_var, surface_geopotential_name, toa_down_solar_input_var, toa_up_solar_var,
                 toa_up_olr_var, surf_down_solar_var, surf_up_solar_var, surf_down_lw_var, surf_up_lw_var, surf_sh_var, surf_lh_var,
                 lead_time_periods, input_source_key="x_physical", **physics):
        super().__init__(input_source_key, physics)
        self.state = [T_var, q_var, U_var, V_var]
        self.sp_var = sp_var
        self.toa_in = toa_down_solar_input_var
        self.toa_pred = [toa_up_solar_var, toa_up_olr_var]
        self.surf = [surf_down_solar_var, surf_up_solar_var, surf_down_lw_var, surf_up_lw_var, surf_sh_var, surf_lh_var]
        self.n_seconds = int(lead_time_periods) * 3600
        self.T_var = T_var

    def _build(self, pb):
        L = self._L
        f0 = 4 * L + 1   # after [T|q|U|V|SP]: SOLIN*dt, USW*dt, OLR*dt, then the six surface terms
        toa = [(f0, 1.0), (f0 + 1, -1.0), (f0 + 2, -1.0)]
        surf = [(f0 + 3, 1.0), (f0 + 4, -1.0), (f0 + 5, 1.0), (f0 + 6, -1.0), (f0 + 7, 1.0), (f0 + 8, 1.0)]
        pb.add_energy_fixer_signed(0, L, 2 * L, 3 * L, toa, surf, np.asarray(self.ph["gph_surf"], np.float32), self.n_seconds)

    def __call__(self, batch_dict):
        ys, xs = self._stack(batch_dict, self.state + [self.sp_var], self.state + [self.sp_var])
        self._L = L = ys[0].shape[1]
        dev = ys[0].device
        ns = float(self.n_seconds)
        ys.append(self._input(batch_dict, self.toa_in)[:, :, -1].to(dev) * ns)       # conservation.py:345-347
        ys += [_pred(batch_dict, k)[:, :, 0] * ns for k in self.toa_pred]
        ys += [_pred(batch_dict, k)[:, :, 0] for k in self.surf]
        y = self._run(torch.cat(ys, 1), torch.cat(xs, 1), 4 * L)
        _set_pred(batch_dict, self.T_var, y[:, :L].unsqueeze(2))
        return batch_dict

    forward = __call__
