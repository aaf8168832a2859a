"""The fixer oracle (oracle/fixers_oracle.py) against the reference's GlobalMass/Water/EnergyFixer on its own
simple_demo grid (tests/golden/fixers_demo.npz, produced by tools/make_goldens.py --only fixers)."""
import os

import numpy as np
import pytest
import torch

from oracle import fixers_oracle as F

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fixers_demo.npz")


def demo_grid(midpoint, dtype=torch.float32):
    lat = np.array([90, 70, 50, 30, 10, -10, -30, -50, -70, -90], dtype=np.float64)
    lon = np.arange(0, 360, 20, dtype=np.float64)
    lon2d, lat2d = np.meshgrid(lon, lat)
    p = np.array([100, 30000, 50000, 70000, 80000, 90000, 100000], dtype=np.float64)
    return F.Grid(lat2d, lon2d, p, midpoint=midpoint, dtype=dtype)


def variant(g, midpoint):
    L = 7
    nl = L - 1 if midpoint else L
    x = np.concatenate([g["x"][b * L:b * L + nl] for b in range(4)], 0)[:, -1]
    y = np.concatenate([g["y"][b * L:b * L + nl] for b in range(4)] + [g["y"][28:]], 0)
    return torch.from_numpy(x), torch.from_numpy(y), nl


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("midpoint", [False, True])
def test_fixers_match_reference(midpoint):
    g = np.load(GOLD)
    tag = "mid" if midpoint else "trapz"
    x, y, nl = variant(g, midpoint)
    grid = demo_grid(midpoint)
    gph = torch.ones(10, 18)
    ns = 6 * 3600.0
    ym = F.mass_fixer(y, x, grid, nl, nl, 3)
    yw = F.water_fixer(y, x, grid, nl, nl, 4 * nl + 6, 4 * nl + 7, ns)
    ye = F.energy_fixer(y, x, grid, 0, nl, 2 * nl, 3 * nl, nl, (4 * nl, 4 * nl + 1), (4 * nl + 2, 4 * nl + 3),
                        (4 * nl + 4, 4 * nl + 5), gph, ns)
    yc = F.energy_fixer(F.water_fixer(F.mass_fixer(y, x, grid, nl, nl, 3), x, grid, nl, nl, 4 * nl + 6, 4 * nl + 7, ns),
                        x, grid, 0, nl, 2 * nl, 3 * nl, nl, (4 * nl, 4 * nl + 1), (4 * nl + 2, 4 * nl + 3),
                        (4 * nl + 4, 4 * nl + 5), gph, ns)
    # per-channel-block relative tolerance: the reference sums ~1e18-magnitude global integrals in fp32
    # (order-dependent at ~1e-6); we sum in fp64
    qs = slice(nl, 2 * nl)
    assert rel(ym[qs].numpy(), g[f"{tag}_mass"][qs]) < 5e-5
    assert rel(yw[4 * nl + 6].numpy(), g[f"{tag}_water"][4 * nl + 6]) < 5e-5
    assert rel(ye[:nl].numpy(), g[f"{tag}_energy"][:nl]) < 5e-5
    for blk, tol in ((slice(0, nl), 1e-4), (qs, 1e-4), (slice(4 * nl + 6, 4 * nl + 7), 2e-3)):
        # chained: the water ratio is a small difference of large global integrals of the (mass-fixed) q, so the
        # 1e-7 fp32 noise of that q shows up at ~2e-4 in the precipitation ratio -- in the reference as well
        assert rel(yc[blk].numpy(), g[f"{tag}_chain"][blk]) < tol
    # untouched channels are bit-identical
    np.testing.assert_array_equal(ym[:nl].numpy(), y[:nl].numpy())
    np.testing.assert_array_equal(yw[:4 * nl + 6].numpy(), y[:4 * nl + 6].numpy())


def test_mass_fixer_conserves_dry_air_mass():
    """The property the reference's gen2 test asserts (tests/test_conservation_gen2.py:241-274): after the fix the
    global dry-air mass of the prediction equals that of the input.  Exact for the midpoint rule; with the
    trapezoidal rule the fixed level ind_fix-1 is also the end point of the last 'held' interval, so the reference
    algorithm itself only conserves to ~1e-4 there."""
    g = np.load(GOLD)
    x, y, nl = variant(g, True)
    grid = demo_grid(True, torch.float64)
    x, y = x.double(), y.double()
    yf = F.mass_fixer(y, x, grid, nl, nl, 3)
    m_in = grid.wsum(F.column_integral(1 - x[nl:2 * nl], grid.p, True) / F.GRAVITY)
    m_out = grid.wsum(F.column_integral(1 - yf[nl:2 * nl], grid.p, True) / F.GRAVITY)
    assert abs(float(m_out - m_in)) <= 1e-9 * abs(float(m_in))
    xt, yt, nt = variant(g, False)
    gt = demo_grid(False, torch.float64)
    yft = F.mass_fixer(yt.double(), xt.double(), gt, nt, nt, 3)
    mi = gt.wsum(F.column_integral(1 - xt[nt:2 * nt].double(), gt.p, False) / F.GRAVITY)
    mo = gt.wsum(F.column_integral(1 - yft[nt:2 * nt], gt.p, False) / F.GRAVITY)
    assert abs(float(mo - mi)) <= 1e-3 * abs(float(mi))


def test_cell_area_matches_torch_gradient():
    lat = torch.tensor([90., 70, 50, 30, 10, -10, -30, -50, -70, -90]).view(-1, 1).expand(10, 18)
    lon = torch.arange(0, 360, 20.).view(1, -1).expand(10, 18)
    d_phi = torch.gradient(torch.sin(torch.deg2rad(lat)), dim=0, edge_order=2)[0]
    d_lam = torch.gradient(torch.deg2rad(lon), dim=1, edge_order=2)[0]
    d_lam = (d_lam + torch.pi) % (2 * torch.pi) - torch.pi
    want = torch.abs(F.RAD_EARTH ** 2 * d_phi * d_lam)
    np.testing.assert_allclose(F.cell_area(lat, lon).numpy(), want.numpy(), rtol=2e-5)  # fp32 edge stencils


# ---- hybrid sigma-pressure grid (tests/golden/fixers_sigma.npz, tools/make_goldens.py --only sigma) -----------------
SIGMA_GOLD = os.path.join(os.path.dirname(__file__), "golden", "fixers_sigma.npz")


def sigma_variant(g, midpoint):
    L = 7
    nl = L - 1 if midpoint else L
    H, W = g["gph"].shape
    x = np.concatenate([g["x"][b * L:b * L + nl] for b in range(4)] + [np.zeros((8, 2, H, W), np.float32), g["sp_x"]], 0)[:, -1]
    y = np.concatenate([g["y"][b * L:b * L + nl] for b in range(4)] + [g["y"][28:], g["sp_y"]], 0)
    return torch.from_numpy(x), torch.from_numpy(y), nl


@pytest.mark.parametrize("midpoint", [False, True])
def test_sigma_fixers_match_reference(midpoint):
    g = np.load(SIGMA_GOLD)
    tag = "mid" if midpoint else "trapz"
    x, y, nl = sigma_variant(g, midpoint)
    lat = np.array([90, 70, 50, 30, 10, -10, -30, -50, -70, -90], dtype=np.float64)
    lon2d, lat2d = np.meshgrid(np.arange(0, 360, 20, dtype=np.float64), lat)
    grid = F.SigmaGrid(lat2d, lon2d, g["coef_a"], g["coef_b"], midpoint=midpoint)
    gph = torch.from_numpy(g["gph"])
    ns, sp = 6 * 3600.0, 4 * nl + 8
    rad = ((4 * nl, 4 * nl + 1), (4 * nl + 2, 4 * nl + 3), (4 * nl + 4, 4 * nl + 5))
    ym = F.mass_fixer_sigma(y, x, grid, nl, nl, sp)
    yw = F.water_fixer_sigma(y, x, grid, nl, nl, 4 * nl + 6, 4 * nl + 7, sp, ns)
    ye = F.energy_fixer_sigma(y, x, grid, 0, nl, 2 * nl, 3 * nl, nl, *rad, sp, gph, ns)
    yc = F.energy_fixer_sigma(F.water_fixer_sigma(F.mass_fixer_sigma(y, x, grid, nl, nl, sp), x, grid, nl, nl, 4 * nl + 6,
                                                  4 * nl + 7, sp, ns), x, grid, 0, nl, 2 * nl, 3 * nl, nl, *rad, sp, gph, ns)
    assert rel(ym[sp].numpy(), g[f"{tag}_mass"][sp]) < 5e-6          # surface pressure rescaled
    np.testing.assert_array_equal(ym[:sp].numpy(), y[:sp].numpy())  # q untouched on sigma grids
    assert rel(yw[4 * nl + 6].numpy(), g[f"{tag}_water"][4 * nl + 6]) < 5e-5
    assert rel(ye[:nl].numpy(), g[f"{tag}_energy"][:nl]) < 5e-5
    for blk, tol in ((slice(0, nl), 1e-4), (slice(sp, sp + 1), 1e-5), (slice(4 * nl + 6, 4 * nl + 7), 2e-3)):
        assert rel(yc[blk].numpy(), g[f"{tag}_chain"][blk]) < tol


UPDOWN_GOLD = os.path.join(os.path.dirname(__file__), "golden", "fixers_updown.npz")


def updown_variant(g, midpoint):
    L = 7
    nl = L - 1 if midpoint else L
    x = np.concatenate([g["x"][b * L:b * L + nl] for b in range(4)], 0)[:, -1]
    y = np.concatenate([g["y"][b * L:b * L + nl] for b in range(4)] + [g["flux"]], 0)
    return torch.from_numpy(x), torch.from_numpy(y), nl


@pytest.mark.parametrize("midpoint", [False, True])
def test_energy_updown_matches_reference(midpoint):
    g = np.load(UPDOWN_GOLD)
    tag = "mid" if midpoint else "trapz"
    x, y, nl = updown_variant(g, midpoint)
    ye = F.energy_fixer_updown(y, x, demo_grid(midpoint), 0, nl, 2 * nl, 3 * nl, nl, [4 * nl + k for k in range(9)],
                               torch.ones(10, 18), 6 * 3600.0)
    assert rel(ye[:nl].numpy(), g[f"{tag}_updown"][:nl]) < 5e-5
    np.testing.assert_array_equal(ye[nl:].numpy(), y[nl:].numpy())
