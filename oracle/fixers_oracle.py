"""CPU ORACLE (test infrastructure) for the conservation fixers of the in-model / outside-model PostBlock
(SURVEY.md §8(a) a12), pressure-level grids.

Restates credit/postblock/gen1.py (GlobalMassFixer :280-391, GlobalWaterFixer :489-569,
GlobalEnergyFixer :704-822) on top of credit/physics_core.py (physics_pressure_level :75-297).
torch CPU; `dtype` selects fp32 (what the engine computes in, with fp64 global sums) or fp64.
Pinned by tests/golden/fixers_demo.npz (the reference classes run on their own `simple_demo` grid,
tools/make_goldens.py --only fixers).

Hybrid sigma-pressure grids (physics_hybrid_sigma_level :300-520; fixer branches gen1.py:306-375, 527-533, 783-788) are
restated by SigmaGrid + the *_sigma functions and pinned by tests/golden/fixers_sigma.npz.

Not reproduced (documented): `concat_fix`'s quirk that DROPS the channels after the
fixed block when it ends at N_vars-2 (gen1.py:1063-1071) — this restatement always keeps every channel.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

GRAVITY = 9.80665      # credit/physics_constants.py
RHO_WATER = 1000.0
LH_WATER = 2.501e6
CP_DRY = 1004.64
CP_VAPOR = 1810.0
RAD_EARTH = 6371000.0


def _gradient_edge2(f: torch.Tensor, dim: int) -> torch.Tensor:
    """torch.gradient(f, dim=dim, edge_order=2) with unit spacing, restated."""
    f = f.movedim(dim, 0)
    g = torch.empty_like(f)
    g[1:-1] = (f[2:] - f[:-2]) / 2
    g[0] = (-3 * f[0] + 4 * f[1] - f[2]) / 2
    g[-1] = (3 * f[-1] - 4 * f[-2] + f[-3]) / 2
    return g.movedim(0, dim)


def cell_area(lat2d: torch.Tensor, lon2d: torch.Tensor) -> torch.Tensor:
    """|R^2 d(sin lat) d(lon)| with the lon difference wrapped into (-pi, pi] (physics_core.py:113-125)."""
    lat_rad = torch.deg2rad(lat2d)
    lon_rad = torch.deg2rad(lon2d)
    d_phi = _gradient_edge2(torch.sin(lat_rad), 0)
    d_lambda = _gradient_edge2(lon_rad, 1)
    d_lambda = (d_lambda + torch.pi) % (2 * torch.pi) - torch.pi
    return torch.abs(RAD_EARTH ** 2 * d_phi * d_lambda)


def column_integral(q: torch.Tensor, p: torch.Tensor, midpoint: bool, a: int = 0, b: Optional[int] = None) -> torch.Tensor:
    """Pressure integral over levels [a, b) of q [L, H, W] (physics_core.py:136-262).

    trapz: sum_{l=a}^{b-2} 0.5 (q_l + q_{l+1}) (p_{l+1} - p_l); midpoint: sum_{l=a}^{b-1} q_l * thickness_l
    where thickness = diff(p) (q then has one level fewer than p)."""
    if midpoint:
        thick = p.diff()
        b = thick.numel() if b is None else b
        return (q[a:b] * thick[a:b].view(-1, 1, 1)).sum(0)
    b = p.numel() if b is None else b
    dp = p[a:b].diff().view(-1, 1, 1)
    qs = q[a:b]
    return (0.5 * (qs[:-1] + qs[1:]) * dp).sum(0)


class Grid:
    def __init__(self, lat2d, lon2d, p_levels, midpoint: bool = False, dtype=torch.float32):
        self.area = cell_area(torch.as_tensor(lat2d).to(dtype), torch.as_tensor(lon2d).to(dtype))
        self.p = torch.as_tensor(p_levels).to(dtype)
        self.midpoint = midpoint
        self.dtype = dtype

    def wsum(self, field: torch.Tensor) -> torch.Tensor:
        return (field.double() * self.area.double()).sum()  # global sums in fp64 (the engine does the same)


def _den(t, mean, std, idx):
    return t if mean is None else t * std[idx].view(-1, 1, 1) + mean[idx].view(-1, 1, 1)


def mass_fixer(y: torch.Tensor, x: torch.Tensor, grid: Grid, q_start: int, n_q: int, fix_level_num: int,
               stats: Optional[Dict] = None) -> torch.Tensor:
    """GlobalMassFixer, pressure levels (gen1.py:300-352).  y [C_out, H, W] (time collapsed), x [C_in, H, W] (last frame)."""
    y = y.clone()
    qi = slice(q_start, q_start + n_q)
    q_in = _den(x[qi], *(stats["in"] if stats else (None, None)), qi) if stats else x[qi]
    q_pr = _den(y[qi], *(stats["out"] if stats else (None, None)), qi) if stats else y[qi]
    n_levels = grid.p.numel()
    ind_fix = n_levels - fix_level_num + 1
    ind_fix_start = ind_fix if grid.midpoint else ind_fix - 1
    m0 = grid.wsum(column_integral(1 - q_in, grid.p, grid.midpoint) / GRAVITY)
    m_hold = grid.wsum(column_integral(1 - q_pr, grid.p, grid.midpoint, 0, ind_fix) / GRAVITY)
    m_fix = grid.wsum(column_integral(1 - q_pr, grid.p, grid.midpoint, ind_fix_start, n_levels) / GRAVITY)
    ratio = ((m0 - m_hold) / m_fix).to(y.dtype)
    q_new = q_pr.clone()
    q_new[ind_fix_start:] = 1 - (1 - q_pr[ind_fix_start:]) * ratio
    if stats:
        mean, std = stats["out"]
        q_new = (q_new - mean[qi].view(-1, 1, 1)) / std[qi].view(-1, 1, 1)
    y[qi] = q_new
    return y


def water_fixer(y: torch.Tensor, x: torch.Tensor, grid: Grid, q_start: int, n_q: int, precip_ind: int, evapor_ind: int,
                n_seconds: float, stats: Optional[Dict] = None) -> torch.Tensor:
    """GlobalWaterFixer, pressure levels (gen1.py:489-569)."""
    y = y.clone()
    qi = slice(q_start, q_start + n_q)
    pi_, ei = slice(precip_ind, precip_ind + 1), slice(evapor_ind, evapor_ind + 1)
    if stats:
        q_in, q_pr = _den(x[qi], *stats["in"], qi), _den(y[qi], *stats["out"], qi)
        precip, evapor = _den(y[pi_], *stats["out"], pi_)[0], _den(y[ei], *stats["out"], ei)[0]
    else:
        q_in, q_pr, precip, evapor = x[qi], y[qi], y[precip_ind], y[evapor_ind]
    twc_in = column_integral(q_in, grid.p, grid.midpoint) / GRAVITY
    twc_pr = column_integral(q_pr, grid.p, grid.midpoint) / GRAVITY
    twc_sum = grid.wsum((twc_pr - twc_in) / n_seconds)
    e_sum = grid.wsum(evapor * RHO_WATER / n_seconds)
    p_sum = grid.wsum(precip * RHO_WATER / n_seconds)
    residual = -twc_sum - e_sum - p_sum
    ratio = ((p_sum + residual) / p_sum).to(y.dtype)
    precip = precip * ratio
    if stats:
        mean, std = stats["out"]
        precip = (precip - mean[precip_ind]) / std[precip_ind]
    y[precip_ind] = precip
    return y


def energy_fixer(y: torch.Tensor, x: torch.Tensor, grid: Grid, T_start: int, q_start: int, U_start: int, V_start: int,
                 n_lev: int, toa_inds: Sequence[int], surf_rad_inds: Sequence[int], surf_flux_inds: Sequence[int],
                 gph_surf: torch.Tensor, n_seconds: float, stats: Optional[Dict] = None) -> torch.Tensor:
    """GlobalEnergyFixer, pressure levels (gen1.py:704-822)."""
    y = y.clone()

    def lev(t, s, which):
        sl = slice(s, s + n_lev)
        return _den(t[sl], *stats[which], sl) if stats else t[sl]

    def one(i):
        sl = slice(i, i + 1)
        return (_den(y[sl], *stats["out"], sl) if stats else y[sl])[0]

    T0, q0, U0, V0 = (lev(x, s, "in") for s in (T_start, q_start, U_start, V_start))
    T1, q1, U1, V1 = (lev(y, s, "out") for s in (T_start, q_start, U_start, V_start))
    cp0 = (1 - q0) * CP_DRY + q0 * CP_VAPOR
    cp1 = (1 - q1) * CP_DRY + q1 * CP_VAPOR
    g = gph_surf.to(y.dtype)
    eq0 = LH_WATER * q0 + g + 0.5 * (U0 ** 2 + V0 ** 2)
    eq1 = LH_WATER * q1 + g + 0.5 * (U1 ** 2 + V1 ** 2)
    r_t = grid.wsum((one(toa_inds[0]) + one(toa_inds[1])) / n_seconds)
    f_s = grid.wsum((one(surf_rad_inds[0]) + one(surf_rad_inds[1]) + one(surf_flux_inds[0]) + one(surf_flux_inds[1])) / n_seconds)
    e0 = cp0 * T0 + eq0
    e1 = cp1 * T1 + eq1
    te0 = grid.wsum(column_integral(e0, grid.p, grid.midpoint) / GRAVITY)
    te1 = grid.wsum(column_integral(e1, grid.p, grid.midpoint) / GRAVITY)
    ratio = ((n_seconds * (r_t - f_s) + te0) / te1).to(y.dtype)
    T_new = (e1 * ratio - eq1) / cp1
    sl = slice(T_start, T_start + n_lev)
    if stats:
        mean, std = stats["out"]
        T_new = (T_new - mean[sl].view(-1, 1, 1)) / std[sl].view(-1, 1, 1)
    y[sl] = T_new
    return y


# --------------------------------------------------------------------------------------------------------------- #
# hybrid sigma-pressure levels: p(l, cell) = a_l + b_l * surface_pressure(cell)   (credit/physics_core.py:300-368)
# --------------------------------------------------------------------------------------------------------------- #
class SigmaGrid:
    def __init__(self, lat2d, lon2d, coef_a, coef_b, midpoint: bool = False, dtype=torch.float32):
        self.area = cell_area(torch.as_tensor(lat2d).to(dtype), torch.as_tensor(lon2d).to(dtype))
        self.a = torch.as_tensor(coef_a).to(dtype)
        self.b = torch.as_tensor(coef_b).to(dtype)
        self.midpoint = midpoint
        self.dtype = dtype

    def wsum(self, field: torch.Tensor) -> torch.Tensor:
        return (field.double() * self.area.double()).sum()

    def integral(self, q: torch.Tensor, sp: torch.Tensor) -> torch.Tensor:
        """pressure_integral_midpoint / _trapz (physics_core.py:369-454): q [L or L-1, H, W], sp [H, W]."""
        pressure = self.a.view(-1, 1, 1) + self.b.view(-1, 1, 1) * sp.unsqueeze(0)
        dp = pressure.diff(dim=0)
        if self.midpoint:
            return (q * dp).sum(0)
        return (0.5 * (q[:-1] + q[1:]) * dp).sum(0)


def mass_fixer_sigma(y, x, grid: SigmaGrid, q_start: int, n_q: int, sp_ind: int, stats: Optional[Dict] = None):
    """GlobalMassFixer, sigma grid (gen1.py:306-313, 355-375): the surface pressure is rescaled, q is untouched."""
    y = y.clone()
    qi = slice(q_start, q_start + n_q)
    si = slice(sp_ind, sp_ind + 1)
    if stats:
        q_in, q_pr = _den(x[qi], *stats["in"], qi), _den(y[qi], *stats["out"], qi)
        sp_in, sp_pr = _den(x[si], *stats["in"], si)[0], _den(y[si], *stats["out"], si)[0]
    else:
        q_in, q_pr, sp_in, sp_pr = x[qi], y[qi], x[sp_ind], y[sp_ind]
    m0 = grid.wsum(grid.integral(1 - q_in, sp_in) / GRAVITY)
    da, db = grid.a.diff().view(-1, 1, 1), grid.b.diff().view(-1, 1, 1)
    dry = (1 - q_pr) if grid.midpoint else 1 - (q_pr[:-1] + q_pr[1:]) / 2
    p_dry_a, p_dry_b = (da * dry).sum(0), (db * dry).sum(0)
    mass_a = grid.wsum(p_dry_a) / GRAVITY
    mass_b = grid.wsum(p_dry_b * sp_pr) / GRAVITY
    ratio = ((m0 - mass_a) / mass_b).to(y.dtype)
    sp_new = sp_pr * ratio
    if stats:
        mean, std = stats["out"]
        sp_new = (sp_new - mean[sp_ind]) / std[sp_ind]
    y[sp_ind] = sp_new
    return y


def water_fixer_sigma(y, x, grid: SigmaGrid, q_start: int, n_q: int, precip_ind: int, evapor_ind: int, sp_ind: int,
                      n_seconds: float, stats: Optional[Dict] = None):
    """GlobalWaterFixer, sigma grid (gen1.py:515-533): column water with each state's own surface pressure."""
    y = y.clone()
    qi = slice(q_start, q_start + n_q)
    pi_, ei, si = slice(precip_ind, precip_ind + 1), slice(evapor_ind, evapor_ind + 1), slice(sp_ind, sp_ind + 1)
    if stats:
        q_in, q_pr = _den(x[qi], *stats["in"], qi), _den(y[qi], *stats["out"], qi)
        precip, evapor = _den(y[pi_], *stats["out"], pi_)[0], _den(y[ei], *stats["out"], ei)[0]
        sp_in, sp_pr = _den(x[si], *stats["in"], si)[0], _den(y[si], *stats["out"], si)[0]
    else:
        q_in, q_pr, precip, evapor, sp_in, sp_pr = x[qi], y[qi], y[precip_ind], y[evapor_ind], x[sp_ind], y[sp_ind]
    twc_in = grid.integral(q_in, sp_in) / GRAVITY
    twc_pr = grid.integral(q_pr, sp_pr) / GRAVITY
    twc_sum = grid.wsum((twc_pr - twc_in) / n_seconds)
    e_sum = grid.wsum(evapor * RHO_WATER / n_seconds)
    p_sum = grid.wsum(precip * RHO_WATER / n_seconds)
    ratio = ((p_sum + (-twc_sum - e_sum - p_sum)) / p_sum).to(y.dtype)
    precip = precip * ratio
    if stats:
        mean, std = stats["out"]
        precip = (precip - mean[precip_ind]) / std[precip_ind]
    y[precip_ind] = precip
    return y


def energy_fixer_sigma(y, x, grid: SigmaGrid, T_start: int, q_start: int, U_start: int, V_start: int, n_lev: int,
                       toa_inds: Sequence[int], surf_rad_inds: Sequence[int], surf_flux_inds: Sequence[int], sp_ind: int,
                       gph_surf: torch.Tensor, n_seconds: float, stats: Optional[Dict] = None):
    """GlobalEnergyFixer, sigma grid (gen1.py:744-788)."""
    y = y.clone()

    def lev(t, s, which):
        sl = slice(s, s + n_lev)
        return _den(t[sl], *stats[which], sl) if stats else t[sl]

    def one(t, i, which):
        sl = slice(i, i + 1)
        return (_den(t[sl], *stats[which], sl) if stats else t[sl])[0]

    T0, q0, U0, V0 = (lev(x, s, "in") for s in (T_start, q_start, U_start, V_start))
    T1, q1, U1, V1 = (lev(y, s, "out") for s in (T_start, q_start, U_start, V_start))
    sp_in, sp_pr = one(x, sp_ind, "in"), one(y, sp_ind, "out")
    cp0 = (1 - q0) * CP_DRY + q0 * CP_VAPOR
    cp1 = (1 - q1) * CP_DRY + q1 * CP_VAPOR
    g = gph_surf.to(y.dtype)
    eq0 = LH_WATER * q0 + g + 0.5 * (U0 ** 2 + V0 ** 2)
    eq1 = LH_WATER * q1 + g + 0.5 * (U1 ** 2 + V1 ** 2)
    r_t = grid.wsum((one(y, toa_inds[0], "out") + one(y, toa_inds[1], "out")) / n_seconds)
    f_s = grid.wsum((one(y, surf_rad_inds[0], "out") + one(y, surf_rad_inds[1], "out") + one(y, surf_flux_inds[0], "out")
                     + one(y, surf_flux_inds[1], "out")) / n_seconds)
    e0 = cp0 * T0 + eq0
    e1 = cp1 * T1 + eq1
    te0 = grid.wsum(grid.integral(e0, sp_in) / GRAVITY)
    te1 = grid.wsum(grid.integral(e1, sp_pr) / GRAVITY)
    ratio = ((n_seconds * (r_t - f_s) + te0) / te1).to(y.dtype)
    T_new = (e1 * ratio - eq1) / cp1
    sl = slice(T_start, T_start + n_lev)
    if stats:
        mean, std = stats["out"]
        T_new = (T_new - mean[sl].view(-1, 1, 1)) / std[sl].view(-1, 1, 1)
    y[sl] = T_new
    return y


def energy_fixer_updown(y, x, grid: Grid, T_start: int, q_start: int, U_start: int, V_start: int, n_lev: int,
                        flux_inds: Sequence[int], gph_surf: torch.Tensor, n_seconds: float, stats: Optional[Dict] = None):
    """GlobalEnergyFixerUpDown, pressure levels (gen1.py:944-1025): the energy fixer with explicit up/down fluxes,
    flux_inds = [TOA down solar, TOA up solar, TOA up OLR, surf down solar, surf up solar, surf down LW, surf up LW, SH, LH]:
    R_T = (d - u - olr) / dt,  F_S = (ds - us + dl - ul - sh - lh) / dt."""
    y = y.clone()

    def lev(t, s, which):
        sl = slice(s, s + n_lev)
        return _den(t[sl], *stats[which], sl) if stats else t[sl]

    def one(i):
        sl = slice(i, i + 1)
        return (_den(y[sl], *stats["out"], sl) if stats else y[sl])[0]

    T0, q0, U0, V0 = (lev(x, s, "in") for s in (T_start, q_start, U_start, V_start))
    T1, q1, U1, V1 = (lev(y, s, "out") for s in (T_start, q_start, U_start, V_start))
    cp0 = (1 - q0) * CP_DRY + q0 * CP_VAPOR
    cp1 = (1 - q1) * CP_DRY + q1 * CP_VAPOR
    g = gph_surf.to(y.dtype)
    eq0 = LH_WATER * q0 + g + 0.5 * (U0 ** 2 + V0 ** 2)
    eq1 = LH_WATER * q1 + g + 0.5 * (U1 ** 2 + V1 ** 2)
    f = [one(i) for i in flux_inds]
    r_t = grid.wsum((f[0] - f[1] - f[2]) / n_seconds)
    f_s = grid.wsum((f[3] - f[4] + f[5] - f[6] - f[7] - f[8]) / n_seconds)
    e0 = cp0 * T0 + eq0
    e1 = cp1 * T1 + eq1
    te0 = grid.wsum(column_integral(e0, grid.p, grid.midpoint) / GRAVITY)
    te1 = grid.wsum(column_integral(e1, grid.p, grid.midpoint) / GRAVITY)
    ratio = ((n_seconds * (r_t - f_s) + te0) / te1).to(y.dtype)
    T_new = (e1 * ratio - eq1) / cp1
    sl = slice(T_start, T_start + n_lev)
    if stats:
        mean, std = stats["out"]
        T_new = (T_new - mean[sl].view(-1, 1, 1)) / std[sl].view(-1, 1, 1)
    y[sl] = T_new
    return y
