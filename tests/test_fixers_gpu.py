"""GPU parity of the device PostBlock (wx_post_* in the C ABI) against the reference's fixers (golden, simple_demo
grid) and the oracle (denorm variant, full-size grid, conservation properties)."""
import os

import numpy as np
import pytest
import torch

from oracle import fixers_oracle as F
from wxengine.engine import WXEngine, WXEngineError, WXPostBlock

from test_fixers_oracle import GOLD, demo_grid, rel, variant

pytestmark = pytest.mark.gpu


def demo_latlon():
    lat = np.array([90, 70, 50, 30, 10, -10, -30, -50, -70, -90], dtype=np.float32)
    lon = np.arange(0, 360, 20, dtype=np.float32)
    lon2d, lat2d = np.meshgrid(lon, lat)
    p = np.array([100, 30000, 50000, 70000, 80000, 90000, 100000], dtype=np.float32)
    return lat2d, lon2d, p


@pytest.mark.parametrize("midpoint", [False, True])
def test_fixers_match_reference_golden(midpoint):
    g = np.load(GOLD)
    tag = "mid" if midpoint else "trapz"
    x, y, nl = variant(g, midpoint)
    lat2d, lon2d, p = demo_latlon()
    ns = 6 * 3600.0
    rad = [4 * nl + k for k in range(6)]
    xd = x[:, None].contiguous().cuda()   # [C_in, frames=1, H, W]

    def run(build):
        pb = WXPostBlock(10, 18, 4 * nl, 1, 4 * nl + 8)
        pb.set_grid(lat2d, lon2d, p, midpoint)
        build(pb)
        yd = y.clone().cuda()
        pb.apply(xd, yd)
        torch.cuda.synchronize()
        return yd.cpu().numpy()

    ym = run(lambda pb: pb.add_mass_fixer(nl, 3))
    yw = run(lambda pb: pb.add_water_fixer(nl, 4 * nl + 6, 4 * nl + 7, ns))
    ye = run(lambda pb: pb.add_energy_fixer(0, nl, 2 * nl, 3 * nl, rad, np.ones((10, 18), np.float32), ns))

    def chain(pb):
        pb.add_mass_fixer(nl, 3)
        pb.add_water_fixer(nl, 4 * nl + 6, 4 * nl + 7, ns)
        pb.add_energy_fixer(0, nl, 2 * nl, 3 * nl, rad, np.ones((10, 18), np.float32), ns)
    yc = run(chain)
    qs = slice(nl, 2 * nl)
    assert rel(ym[qs], g[f"{tag}_mass"][qs]) < 5e-5
    assert rel(yw[4 * nl + 6], g[f"{tag}_water"][4 * nl + 6]) < 5e-5
    assert rel(ye[:nl], g[f"{tag}_energy"][:nl]) < 5e-5
    for blk, tol in ((slice(0, nl), 1e-4), (qs, 1e-4), (slice(4 * nl + 6, 4 * nl + 7), 2e-3)):
        assert rel(yc[blk], g[f"{tag}_chain"][blk]) < tol
    # channels a fixer does not own are untouched, bit for bit
    np.testing.assert_array_equal(ym[:nl], y[:nl].numpy())
    np.testing.assert_array_equal(yw[:4 * nl + 6], y[:4 * nl + 6].numpy())
    np.testing.assert_array_equal(ye[nl:], y[nl:].numpy())


def test_full_size_denorm_vs_oracle_and_conservation():
    """721x1440, 13 levels, denorm=True: against the fp64-sum oracle, plus the conservation property itself."""
    H, W, L = 721, 1440, 13
    rng = np.random.Generator(np.random.Philox(key=[5, 5]))
    lat = np.linspace(90, -90, H, dtype=np.float32)
    lon = np.arange(W, dtype=np.float32) * 0.25
    lon2d, lat2d = np.meshgrid(lon, lat)
    p = np.array([5000, 10000, 15000, 20000, 25000, 30000, 40000, 50000, 60000, 70000, 85000, 92500, 100000], np.float32)
    c_in, c_out = 4 * L, 4 * L + 8
    mean_in = rng.standard_normal(c_in).astype(np.float32) * 0.0
    std_in = np.ones(c_in, np.float32)
    # physical-ish fields expressed in NORMALISED space: q block has mean 0.004 / std 0.003 etc.
    mean_out = np.zeros(c_out, np.float32); std_out = np.ones(c_out, np.float32)
    mean_out[:L] = 250; std_out[:L] = 30; mean_out[L:2 * L] = 0.004; std_out[L:2 * L] = 0.003
    std_out[2 * L:4 * L] = 10; std_out[4 * L:4 * L + 6] = 2e6; mean_out[4 * L + 6] = 2e-3; std_out[4 * L + 6] = 1e-3
    mean_out[4 * L + 7] = -1e-3; std_out[4 * L + 7] = 5e-4
    mean_in[:] = mean_out[:c_in]; std_in[:] = std_out[:c_in]
    x = torch.from_numpy(rng.standard_normal((c_in, 1, H, W), dtype=np.float32) * 0.5)
    y = torch.from_numpy(rng.standard_normal((c_out, H, W), dtype=np.float32) * 0.5)
    pb = WXPostBlock(H, W, c_in, 1, c_out)
    pb.set_grid(lat2d, lon2d, p, True)
    pb.set_stats(mean_in, std_in, mean_out, std_out)
    pb.add_mass_fixer(L, 3, denorm=True)
    yd = y.clone().cuda()
    pb.apply(x.cuda(), yd)
    got = yd.cpu()
    nl = L - 1
    grid = F.Grid(lat2d, lon2d, p, midpoint=True)
    stats = {"in": (torch.from_numpy(mean_in), torch.from_numpy(std_in)), "out": (torch.from_numpy(mean_out), torch.from_numpy(std_out))}
    ref = F.mass_fixer(y, x[:, 0], grid, L, nl, 3, stats)
    assert float((got[L:L + nl] - ref[L:L + nl]).abs().max()) < 5e-4   # normalised units; q_phys error ~1e-7/std
    # conservation (midpoint rule: exact up to fp32): dry-air mass after == before
    q_in = (x[L:L + nl, 0] * 0.003 + 0.004).double()
    q_out = (got[L:L + nl] * 0.003 + 0.004).double()
    g64 = F.Grid(lat2d, lon2d, p, midpoint=True, dtype=torch.float64)
    m_in = g64.wsum(F.column_integral(1 - q_in, g64.p, True) / F.GRAVITY)
    m_out = g64.wsum(F.column_integral(1 - q_out, g64.p, True) / F.GRAVITY)
    assert abs(float(m_out - m_in)) < 2e-6 * abs(float(m_in))


def test_attached_postblock_runs_inside_the_model_step():
    """In-model use (PostBlock inside CrossFormer.forward, crossformer.py:637-642): attached fixers run after the
    forward and before y_phys / x_next are formed."""
    from wxengine.config import named_config
    from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict
    cfg = named_config("T0")
    eng = WXEngine(cfg, "fp32", 0)
    eng.load_state_dict(synth_state_dict(cfg))
    eng.finalize()
    n_prog = cfg.channels * cfg.levels + cfg.surface_channels
    mean, std = synth_denorm(cfg.base_output_channels)
    eng.set_denorm(mean, std)
    eng.set_layout(n_prog, 2, 2)
    x = torch.from_numpy(synth_input(cfg)).cuda()
    y_plain = eng.forward(x).clone()
    H, W = cfg.image_height, cfg.image_width
    pb = WXPostBlock(H, W, cfg.base_input_channels, 1, cfg.base_output_channels)
    lat = np.linspace(88, -88, H, dtype=np.float32); lon = np.arange(W, dtype=np.float32) * (360.0 / W)
    lon2d, lat2d = np.meshgrid(lon, lat)
    pb.set_grid(lat2d, lon2d, np.array([30000, 60000, 100000], np.float32), False)
    pb.add_mass_fixer(3 * cfg.levels, 2)   # q block = 4th 3-D variable
    eng.attach_postblock(pb)
    frc = torch.from_numpy(synth_forcing(cfg, 2, 1)).cuda()
    y, yp, xn = eng.step(x, frc)
    ref = pb.apply(x[0], y_plain[0, :, 0].clone())          # same op applied outside the model
    assert torch.equal(y[0, :, 0], ref)
    assert not torch.equal(y, y_plain)
    np.testing.assert_allclose(yp[0].cpu().numpy(), y[0, :, 0].cpu().numpy() * std[:, None, None] + mean[:, None, None], rtol=1e-6, atol=1e-6)
    assert torch.equal(xn[0, :n_prog, 0], y[0, :n_prog, 0])
    eng.attach_postblock(None)
    assert torch.equal(eng.forward(x), y_plain)
    with pytest.raises(WXEngineError):
        eng.attach_postblock(WXPostBlock(10, 18, 4, 1, 4))   # geometry mismatch


def test_post_errors():
    pb = WXPostBlock(10, 18, 28, 1, 36)
    with pytest.raises(WXEngineError, match="set_grid"):
        pb.add_mass_fixer(7, 3)
    lat2d, lon2d, p = demo_latlon()
    pb.set_grid(lat2d, lon2d, p, False)
    with pytest.raises(WXEngineError, match="set_stats"):
        pb.add_mass_fixer(7, 3, denorm=True)
    with pytest.raises(WXEngineError, match="out of range"):
        pb.add_water_fixer(7, 99, 35, 21600.0)


@pytest.mark.parametrize("midpoint", [False, True])
def test_sigma_fixers_match_reference_golden(midpoint):
    """Hybrid sigma-pressure grid (wx_post_set_grid_sigma) against the reference's sigma branches (fixers_sigma.npz)."""
    from test_fixers_oracle import SIGMA_GOLD, sigma_variant
    g = np.load(SIGMA_GOLD)
    tag = "mid" if midpoint else "trapz"
    x, y, nl = sigma_variant(g, midpoint)
    lat2d, lon2d, _ = demo_latlon()
    ns, sp = 6 * 3600.0, 4 * nl + 8
    rad = [4 * nl + k for k in range(6)]
    xd = x[:, None].contiguous().cuda()

    def run(build):
        pb = WXPostBlock(10, 18, x.shape[0], 1, y.shape[0])
        pb.set_grid_sigma(lat2d, lon2d, g["coef_a"], g["coef_b"], sp, midpoint)
        build(pb)
        yd = y.clone().cuda()
        pb.apply(xd, yd)
        torch.cuda.synchronize()
        return yd.cpu().numpy()

    ym = run(lambda pb: pb.add_mass_fixer(nl, 3))
    yw = run(lambda pb: pb.add_water_fixer(nl, 4 * nl + 6, 4 * nl + 7, ns))
    ye = run(lambda pb: pb.add_energy_fixer(0, nl, 2 * nl, 3 * nl, rad, g["gph"], ns))

    def chain(pb):
        pb.add_mass_fixer(nl, 3)
        pb.add_water_fixer(nl, 4 * nl + 6, 4 * nl + 7, ns)
        pb.add_energy_fixer(0, nl, 2 * nl, 3 * nl, rad, g["gph"], ns)
    yc = run(chain)
    assert rel(ym[sp], g[f"{tag}_mass"][sp]) < 5e-6
    np.testing.assert_array_equal(ym[:sp], y[:sp].numpy())   # on sigma grids the mass fixer leaves q alone
    assert rel(yw[4 * nl + 6], g[f"{tag}_water"][4 * nl + 6]) < 5e-5
    assert rel(ye[:nl], g[f"{tag}_energy"][:nl]) < 5e-5
    for blk, tol in ((slice(0, nl), 1e-4), (slice(sp, sp + 1), 1e-5), (slice(4 * nl + 6, 4 * nl + 7), 2e-3)):
        assert rel(yc[blk], g[f"{tag}_chain"][blk]) < tol
    with pytest.raises(WXEngineError):  # the surface-pressure channel must exist
        WXPostBlock(10, 18, x.shape[0], 1, y.shape[0]).set_grid_sigma(lat2d, lon2d, g["coef_a"], g["coef_b"], 999, midpoint)


@pytest.mark.parametrize("midpoint", [False, True])
def test_energy_updown_matches_reference_golden(midpoint):
    from test_fixers_oracle import UPDOWN_GOLD, updown_variant
    g = np.load(UPDOWN_GOLD)
    tag = "mid" if midpoint else "trapz"
    x, y, nl = updown_variant(g, midpoint)
    lat2d, lon2d, p = demo_latlon()
    pb = WXPostBlock(10, 18, 4 * nl, 1, 4 * nl + 9)
    pb.set_grid(lat2d, lon2d, p, midpoint)
    pb.add_energy_fixer_updown(0, nl, 2 * nl, 3 * nl, [4 * nl + k for k in range(9)], np.ones((10, 18), np.float32), 6 * 3600.0)
    yd = y.clone().cuda()
    pb.apply(x[:, None].contiguous().cuda(), yd)
    torch.cuda.synchronize()
    ye = yd.cpu().numpy()
    assert rel(ye[:nl], g[f"{tag}_updown"][:nl]) < 5e-5
    np.testing.assert_array_equal(ye[nl:], y[nl:].numpy())
