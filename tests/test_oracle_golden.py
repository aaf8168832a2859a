"""The oracle (oracle/wxformer_oracle.py) against golden outputs of the REAL reference.

Fixtures were produced by tools/make_goldens.py (reference imported from /root/reference,
CPU fp32, synthetic name-keyed weights).  Tolerance: both sides are fp32 evaluations of the
same function with different op orders; the reference's own sharded-vs-unsharded gates use
atol 1e-5 per layer (tests/test_domain_parallel_multigpu.py:115) — we allow 2e-5*max|y| end to end.
"""
import os

import numpy as np
import pytest
import torch

from oracle import wxformer_oracle as O
from wxengine.config import named_config
from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def test_earth_pad_small_exact():
    g = _load("earth_pad.npz")
    x = torch.from_numpy(g["small_x"])
    got = O.earth_pad(x, (3, 2), (4, 3)).numpy()
    np.testing.assert_array_equal(got, g["small_pad"])
    back = O.earth_unpad(torch.from_numpy(got), (3, 2), (4, 3)).numpy()
    np.testing.assert_array_equal(back, g["small_x"])


def test_earth_pad_asymmetric_181x360():
    g = _load("earth_pad.npz")
    rng = np.random.Generator(np.random.Philox(key=[7, 7]))
    rng.standard_normal((1, 2, 1, 7, 10), dtype=np.float32)  # consume the 'small' draw
    big = torch.from_numpy(rng.standard_normal((1, 3, 1, 181, 360), dtype=np.float32))
    pb = O.earth_pad(big, (12, 34), (56, 78))
    assert list(pb.shape) == list(g["big_pad_shape"])
    np.testing.assert_array_equal(pb[0, :, 0, ::7, ::11].numpy(), g["big_pad_strided"])
    np.testing.assert_allclose(pb.double().sum(dim=(0, 2, 3, 4)).numpy(), g["big_pad_sum"], rtol=1e-12)


def test_mirror_pad_matches_reference():
    """padding mode "mirror" (credit/boundary_padding.py:98-134): reflect in latitude, wrap in longitude."""
    g = _load("earth_pad.npz")
    x = torch.from_numpy(g["small_x"])
    np.testing.assert_array_equal(O.mirror_pad(x, (3, 2), (4, 3)).numpy(), g["small_mirror"])
    rng = np.random.Generator(np.random.Philox(key=[7, 7]))
    rng.standard_normal((1, 2, 1, 7, 10), dtype=np.float32)
    big = torch.from_numpy(rng.standard_normal((1, 3, 1, 181, 360), dtype=np.float32))
    pm = O.mirror_pad(big, (12, 34), (56, 78))
    np.testing.assert_array_equal(pm[0, :, 0, ::7, ::11].numpy(), g["big_mirror_strided"])
    np.testing.assert_allclose(pm.double().sum(dim=(0, 2, 3, 4)).numpy(), g["big_mirror_sum"], rtol=1e-12)
    np.testing.assert_array_equal(O.earth_unpad(pm, (12, 34), (56, 78)).numpy(), big.numpy())


_SLOW = os.environ.get("WX_SLOW", "0") == "1"


@pytest.mark.parametrize("family", ["stress", "stress_hi"])
@pytest.mark.parametrize("name", ["T0", "T1", "C1"])
def test_stress_weight_families_match_reference_golden(name, family):
    """The stress weight families of wxengine.synth (logits of +-40, pre-GELU 1e2, LayerNorm / GroupNorm inputs with |mean| / sigma of
    100-240, a hidden unit beyond the f16 range) through the REAL reference (tools/make_goldens.py --only stress) vs the oracle."""
    g = _load(f"model_{name}_{family}.npz")
    cfg = named_config(name)
    y = O.forward(cfg, synth_state_dict(cfg, family=family), synth_input(cfg))
    s = int(g["stride"])
    scale = float(np.abs(g["y"]).max())
    assert np.abs(y[0, :, 0, ::s, ::s].numpy() - g["y"]).max() <= 2e-5 * scale


def test_stress_families_differ_from_base_only_where_documented():
    cfg = named_config("T0")
    base, st, hi = (synth_state_dict(cfg, family=f) for f in ("base", "stress", "stress_hi"))
    assert list(base) == list(st) == list(hi)
    changed = {k for k in base if not np.array_equal(base[k], st[k])}
    assert changed and all(k.endswith((".bias", ".g", ".weight_u", ".weight_v")) or ".0.convs." in k for k in changed)
    k1 = "layers.0.1.layers.0.1.layers.1.bias"
    assert hi[k1][3] == 7.0e4 and st[k1][3] == base[k1][3]


@pytest.mark.parametrize("name", ["T0", "T1", "C1", "C3S", "T0W", "C1W", "T0U", "T0M", "T0F", "RT", "T0H", "T1H", "T0X",
                                  pytest.param("C3", marks=pytest.mark.skipif(
                                      not _SLOW, reason="~1.5 min of CPU; set WX_SLOW=1"))])
def test_forward_matches_reference_golden(name):
    g = _load(f"model_{name}.npz")
    cfg = named_config(name)
    sd = synth_state_dict(cfg)
    cap = {} if name in ("T0", "T0W", "T0U") else None
    y = O.forward(cfg, sd, synth_input(cfg), capture=cap)
    s = int(g["stride"])
    ys = y[0, :, 0, ::s, ::s].numpy()
    scale = float(np.abs(g["y"]).max())
    assert np.abs(ys - g["y"]).max() <= 2e-5 * scale
    a = y[0, :, 0].double()
    np.testing.assert_allclose(a.sum(dim=(1, 2)).numpy(), g["ch_sum"], rtol=0, atol=2e-5 * a.shape[1] * a.shape[2])
    np.testing.assert_allclose((a * a).sum(dim=(1, 2)).numpy(), g["ch_sumsq"], rtol=1e-4)
    if cap is not None:  # layer-by-layer pins
        for key in g.files:
            if not key.startswith("cap/"):
                continue
            ref = g[key]
            got = cap[key[4:]][0].numpy()
            assert got.shape == ref.shape, key
            assert np.abs(got - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), key


def test_rollout_glue_matches_reference_golden():
    """model -> TracerFixer -> denorm -> update_x, 3 steps, against the reference's own pieces."""
    g = _load("rollout_T0.npz")
    cfg = named_config("T0")
    sd = synth_state_dict(cfg)
    mean, std = synth_denorm(cfg.base_output_channels)
    frc = [synth_forcing(cfg, int(g["n_dyn"]), t) for t in (1, 2, 3)]
    tracer = dict(inds=[int(i) for i in g["tracer_inds"]], thres=[float(t) for t in g["tracer_thres"]], denorm=False)
    ys, phys = O.rollout(cfg, sd, synth_input(cfg), frc, int(g["n_static"]), mean, std, tracer)
    x = torch.from_numpy(synth_input(cfg))
    n_prog = cfg.channels * cfg.levels + cfg.surface_channels
    for t in (1, 2, 3):
        tol = 3e-5 * t  # error feeds back through the autoregressive loop
        assert np.abs(ys[t - 1][0, :, 0].numpy() - g[f"y{t}"]).max() <= tol * 2.0
        assert np.abs(phys[t - 1][0].numpy() - g[f"yphys{t}"]).max() <= tol * 4.0
        x = O.update_x(x, torch.from_numpy(frc[t - 1]), ys[t - 1], n_prog, int(g["n_static"]))
        assert np.abs(x[0, :, 0].numpy() - g[f"x{t}"]).max() <= tol * 2.0
        # clamped cells are exactly the threshold
        q = ys[t - 1][0, tracer["inds"], 0]
        assert float(q.min()) >= tracer["thres"][0] - 1e-7


def test_two_source_channel_layout_matches_reference_golden():
    """build_channel_layout / update_x with interleaving sources (channel_utils.py:161-291): the host mirror reproduces the
    reference's groups, the oracle's group-wise update reproduces its x_new bit for bit."""
    from synth_batches import two_source_conf
    from wxengine.rollout import build_channel_layout
    g = _load("channel_layout_two_sources.npz")
    groups, n_pred = build_channel_layout(two_source_conf())
    code = {"prognostic": 0, "dynamic_forcing": 1, "static": 2}
    mine = [(code[k], x0, -1 if s0 is None else s0, n) for k, x0, s0, n in groups]
    assert mine == [tuple(int(v) for v in row) for row in g["groups"]]
    assert n_pred == int(g["n_pred"]) == 9
    xn = O.update_x_groups(torch.from_numpy(g["x"]), torch.from_numpy(g["frc"]), torch.from_numpy(g["y"]), groups)
    np.testing.assert_array_equal(xn.numpy(), g["x_new"])
    with pytest.raises(ValueError, match="history_len"):
        bad = two_source_conf()
        bad["data"]["source"]["aux"]["history_len"] = 2
        build_channel_layout(bad)


def test_tracer_fix_denorm_roundtrip():
    """denorm: True branch (gen1.py:147-161): clamp happens in physical units."""
    torch.manual_seed(1)
    y = torch.randn(1, 6, 1, 5, 7)
    mean = torch.linspace(-1, 1, 6)
    std = torch.linspace(0.5, 2, 6)
    out = O.tracer_fix(y, [2, 4], [0.1, -0.2], mean, std)
    phys = out * std.view(1, -1, 1, 1, 1) + mean.view(1, -1, 1, 1, 1)
    assert float(phys[:, 2].min()) >= 0.1 - 1e-6 and float(phys[:, 4].min()) >= -0.2 - 1e-6
    untouched = [0, 1, 3, 5]
    np.testing.assert_allclose(out[:, untouched].numpy(), y[:, untouched].numpy(), atol=1e-6)


def test_bilinear_matches_torch():
    torch.manual_seed(0)
    x = torch.randn(1, 3, 36, 72)
    np.testing.assert_allclose(O.bilinear_resize(x, 37, 72).numpy(),
                               torch.nn.functional.interpolate(x, size=(37, 72), mode="bilinear").numpy(), atol=1e-6)
    np.testing.assert_allclose(O.bilinear_resize(x, 41, 80).numpy(),
                               torch.nn.functional.interpolate(x, size=(41, 80), mode="bilinear").numpy(), atol=2e-6)
    x = torch.randn(1, 2, 720, 360)
    np.testing.assert_allclose(O.bilinear_resize(x, 721, 360).numpy(),
                               torch.nn.functional.interpolate(x, size=(721, 360), mode="bilinear").numpy(), atol=2e-6)


def test_window_partition_roundtrip_and_long_stride():
    c, h, w, wsz = 2, 12, 24, 4
    x = torch.arange(c * h * w, dtype=torch.float32).reshape(c, h, w)
    for kind in ("short", "long"):
        t = O.window_partition(x, wsz, kind)
        assert t.shape == ((h // wsz) * (w // wsz), wsz * wsz, c)
        assert torch.equal(O.window_merge(t, h, w, wsz, kind), x)
    t = O.window_partition(x, wsz, "long")
    # window (hh=1, ww=2), token (l1=3, l2=1) sits at pixel (3*(h/wsz)+1, 1*(w/wsz)+2)
    win = 1 * (w // wsz) + 2
    assert float(t[win, 3 * wsz + 1, 0]) == float(x[0, 3 * (h // wsz) + 1, 1 * (w // wsz) + 2])
