"""GPU parity tests proper: the HIP engine (through the C ABI) against the CPU oracle, the committed
reference goldens, and size-independent properties at BASELINE.json's full sizes.

Tolerances (stated):
  fp32 engine (exact-f32 MFMA):  max|y - ref| <= 1e-4 * max|ref|  (observed ~2e-6)
  bf16 engine (bf16 storage/MFMA, fp32 accumulate): rel-L2 <= 2e-2, max err <= 5e-2*max|ref| (observed ~8e-3)
"""
import os

import numpy as np
import pytest
import torch

from oracle import wxformer_oracle as O
from wxengine.config import named_config
from wxengine.engine import WXEngine, WXEngineError
from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FP32_TOL = 1e-4
BF16_L2, BF16_MAX = 2e-2, 5e-2
_engines = {}


def get_engine(name, prec):
    key = (name, prec)
    if key not in _engines:
        cfg = named_config(name)
        eng = WXEngine(cfg, prec, 0)
        eng.load_state_dict(synth_state_dict(cfg))
        eng.finalize()
        _engines[key] = eng
    return _engines[key]


def check(y, ref, prec):
    y, ref = np.asarray(y, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = np.abs(ref).max()
    err = np.abs(y - ref).max()
    assert np.isfinite(y).all()
    if prec in ("fp32", "fp32s"):   # fp32s: fp32 storage + split-bf16 GEMM arithmetic -- the SAME stated tolerance
        assert err <= FP32_TOL * scale, f"{prec} max err {err:.3e} vs scale {scale:.3e}"
    else:
        l2 = np.linalg.norm(y - ref) / np.linalg.norm(ref)
        assert l2 <= BF16_L2 and err <= BF16_MAX * scale, f"bf16 rel-L2 {l2:.3e} max err {err:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
@pytest.mark.parametrize("name", ["T0", "T1", "T0W", "T0U", "T0M", "T0F", "T0H", "T1H", "T0X"])
def test_forward_and_every_block_vs_oracle(name, prec):
    cfg = named_config(name)
    sd = synth_state_dict(cfg)
    x = synth_input(cfg)
    cap = {}
    y_ref = O.forward(cfg, sd, x, capture=cap)
    eng = get_engine(name, prec)
    eng.set_debug(True)
    y = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    check(y, y_ref.numpy(), prec)
    n = 0
    for k, v in cap.items():
        got = eng.debug_read(k)
        assert got.shape == tuple(v.shape[1:]), k
        if k == "pad" and prec in ("fp32", "fp32s"):
            np.testing.assert_array_equal(got, v[0].numpy())  # pure data movement: bit exact
        else:
            check(got, v[0].numpy(), prec)
        n += 1
    eng.set_debug(False)
    assert n >= 20


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
@pytest.mark.parametrize("name", ["C1", "C1W", "RT"])
def test_c1_vs_oracle_and_reference_golden(name, prec):
    """1-degree configs of both reference classes (legacy ConvTranspose decoder, wxformer PixelShuffle decoder), and RT = the
    model of the reference's own unit test (tests/test_crossformer.py): 256-token long windows, three CrossEmbed kernels,
    pads of half the image, upsample_v_conv decoder."""
    cfg = named_config(name)
    x = synth_input(cfg)
    y = get_engine(name, prec).forward(torch.from_numpy(x).cuda()).cpu()
    y_ref = O.forward(cfg, synth_state_dict(cfg), x)
    check(y.numpy(), y_ref.numpy(), prec)
    g = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    s = int(g["stride"])
    check(y[0, :, 0, ::s, ::s].numpy(), g["y"], prec)


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
@pytest.mark.parametrize("name", ["T0H", "T1H", "T0X"])
def test_wide_heads_vs_reference_golden(name, prec):
    """dim_head = 64 / 128 (crossformer.py:372-401; the reference's YAMLs leave the default 32): the general-head-dimension attention
    kernel between the plain GEMMs, against the reference's own fp32 forward (tools/make_goldens.py --only T0H / T1H), same gates
    as the 32-wide heads.  The per-block oracle captures of these configs are in test_forward_and_every_block_vs_oracle."""
    cfg = named_config(name)
    y = get_engine(name, prec).forward(torch.from_numpy(synth_input(cfg)).cuda()).cpu()
    g = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    s = int(g["stride"])
    check(y[0, :, 0, ::s, ::s].numpy(), g["y"], prec)


def _stress_engine(name, prec, family):
    cfg = named_config(name)
    eng = WXEngine(cfg, prec, 0)
    eng.load_state_dict(synth_state_dict(cfg, family=family))
    eng.finalize()
    return cfg, eng


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
@pytest.mark.parametrize("name", ["T0", "T1", "C1", "C3S", "C3"])
def test_stress_weights_vs_reference_golden(name, prec):
    """Weight family "stress" (wxengine.synth.FAMILIES): softmax logits of +-40 and more, FeedForward pre-GELU magnitudes of ~1e2,
    one-sign conv biases in front of the first LayerNorm and of every second GroupNorm.  Golden = the reference's fp32 CPU forward
    (tools/make_goldens.py --only stress).  fp32 engine: the SAME gate as the base family, 1e-4 * max|y| (measured 4e-6 .. 1.1e-5).
    bf16 engine: with LayerNorm gains of 8 / 60 every sub-block's update dwarfs the stream it is added to, so bf16 rounding is renewed,
    not damped, layer by layer: the REFERENCE ITSELF under torch.autocast(bfloat16) sits 3.1e-2 .. 3.5e-2 (rel-L2) from its own fp32
    output on this family (0.84e-2 .. 0.90e-2 on the base family; stored in the fixture as bf16_autocast_l2).  Gate: the engine must be
    at least as close as that, and inside 3e-2 (measured 1.9e-2 .. 2.4e-2; base family: 0.75e-2 .. 0.81e-2 under the 2e-2 gate)."""
    cfg, eng = _stress_engine(name, prec, "stress")
    y = eng.forward(torch.from_numpy(synth_input(cfg)).cuda()).cpu()
    g = np.load(os.path.join(GOLD, f"model_{name}_stress.npz"))
    s = int(g["stride"])
    ys = y[0, :, 0, ::s, ::s].numpy()
    if prec in ("fp32", "fp32s"):
        check(ys, g["y"], prec)
        return
    assert np.isfinite(ys).all()
    l2 = np.linalg.norm(ys.astype(np.float64) - g["y"]) / np.linalg.norm(g["y"])
    print(f"stress {name} bf16 rel-L2 {l2:.3e} (reference under bf16 autocast: {float(g['bf16_autocast_l2']):.3e})")
    assert l2 <= 3e-2 and l2 <= float(g["bf16_autocast_l2"])


@pytest.mark.parametrize("prec", ["fp32", "fp32s"])
@pytest.mark.parametrize("name", ["T0", "T1", "C1", "C3S", "C3"])
def test_stress_hi_fp32_vs_reference_golden(name, prec):
    """Weight family "stress_hi": LayerNorm rows (stage 0) and GroupNorm groups with |mean| / sigma of 100-240 -- where a one-pass
    variance sum(x^2)/C - mean^2 cancels -- and one FeedForward hidden unit per layer at 7e4.  fp32 engine, the stated fp32 gate
    (round 5: also at the headline map size, C3S / C3, and for the split-bf16 mode -- whose LayerNorm is applied to the operand
    before the split, so no mean * colsum cancellation meets its 2^-17 product error)."""
    cfg, eng = _stress_engine(name, prec, "stress_hi")
    y = eng.forward(torch.from_numpy(synth_input(cfg)).cuda()).cpu()
    g = np.load(os.path.join(GOLD, f"model_{name}_stress_hi.npz"))
    s = int(g["stride"])
    check(y[0, :, 0, ::s, ::s].numpy(), g["y"], prec)


@pytest.mark.parametrize("name", ["T0", "T1", "C1", "C3S", "C3"])
def test_stress_hi_bf16_stays_finite_and_bounded(name, monkeypatch):
    """The same family on the bf16 engine.  A bf16 residual stream at |mean| / sigma = r carries r * 2^-8 / sqrt(12) of rounding noise
    per normalised element BEFORE any kernel touches it (r = 100-200 here: 10-20 %), so the 2e-2 gate is out of reach by construction of
    the storage format, not of a kernel; what must hold: finite output (the 7e4 hidden unit is beyond f16, which the fused FeedForward
    kernels use for their hidden activations -- they saturate instead of producing inf * 0), and an error bounded by that noise."""
    monkeypatch.setenv("WX_FF_MIN_WGS", "0")   # force the fused (f16-hidden) FeedForward kernels onto these small maps
    cfg, eng = _stress_engine(name, "bf16", "stress_hi")
    y = eng.forward(torch.from_numpy(synth_input(cfg)).cuda()).cpu()
    assert torch.isfinite(y).all()
    g = np.load(os.path.join(GOLD, f"model_{name}_stress_hi.npz"))
    s = int(g["stride"])
    ys = y[0, :, 0, ::s, ::s].numpy().astype(np.float64)
    l2 = np.linalg.norm(ys - g["y"]) / np.linalg.norm(g["y"])
    print(f"stress_hi {name} bf16 rel-L2 {l2:.3e} (reference under bf16 autocast: {float(g['bf16_autocast_l2']):.3e})")
    # measured 4.4e-3 .. 5.6e-3 (the reference under torch.autocast(bf16): 4.9e-3 .. 5.7e-3): y is dominated by the decoder's one-sign
    # biases here, so the END-TO-END metric does not see the stage-0 stream noise (tools/stress_report.py prints it layer by layer)
    assert l2 <= 2e-2


def test_step_with_two_interleaved_sources():
    """wx_set_layout_groups: the next input is assembled per (source, field type) group -- prognostic channels of two sources
    come from non-adjacent blocks of y (each source's diagnostics sit in between), forcing channels from one forcing tensor,
    statics are carried.  Copies are exact; the prognostic channels equal the step's own y."""
    from synth_batches import two_source_conf
    from wxengine.config import WXConfig
    from wxengine.latband import VirtualBands
    from wxengine.rollout import build_channel_layout
    mc = dict(frames=1, channels=2, surface_channels=3, input_only_channels=5, output_only_channels=2, levels=3,
              image_height=37, image_width=72, patch_width=1, patch_height=1, cross_embed_kernel_sizes=[[4, 8, 16, 32], [2, 4], [2, 4], [2, 4]],
              cross_embed_strides=[2, 2, 2, 2], dim=[32, 64, 128, 256], depth=[1, 1, 1, 1], global_window_size=[4, 2, 2, 1],
              local_window_size=3, interp=True, use_spectral_norm=True,
              padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]))
    cfg = WXConfig.from_model_conf(mc)
    groups, n_pred = build_channel_layout(two_source_conf())
    assert cfg.base_input_channels == 14 and cfg.base_output_channels == 11 and n_pred == 9
    sd = synth_state_dict(cfg)
    eng = WXEngine(cfg, "fp32", 0)
    eng.load_state_dict(sd)
    eng.finalize()
    eng.set_layout_groups(groups)
    x = torch.from_numpy(synth_input(cfg)).cuda()
    frc = torch.from_numpy(synth_forcing(cfg, 3, 1)).cuda()
    y, _, xn = eng.step(x, frc, want_phys=False)
    want = O.update_x_groups(x.cpu(), frc.cpu(), y.cpu(), groups)
    assert torch.equal(xn.cpu(), want)
    # the same layout on three lat-band ranks
    vb = VirtualBands(cfg, sd, 3, "fp32", setup=lambda e: e.set_layout_groups(groups))
    ys, _, xs = vb.step(x, frc, want_next=True)
    assert torch.equal(xs.cpu(), O.update_x_groups(x.cpu(), frc.cpu(), ys.cpu(), groups))
    assert (ys - y).abs().max().item() <= 1e-5 * y.abs().max().item()
    with pytest.raises(WXEngineError, match="cover every input channel"):
        eng.set_layout_groups(groups[:-1])
    with pytest.raises(WXEngineError, match="two groups"):
        eng.set_layout_groups(groups + [("static", 0, None, 1)])


@pytest.mark.parametrize("name", ["T1", "C1"])
def test_fused_feed_forward_kernel_on_small_maps(name, monkeypatch):
    """By default the fused feed-forward kernel (wx_ff.h) only runs where it yields >= 256 workgroups (0.25-degree stages 0/1; >= 40 at C = 128);
    WX_FF_MIN_WGS=0 forces it onto the small maps so that its every variant (plain, +out-proj, +out-proj+qkv) is also checked
    against the oracle at sizes the oracle finishes in seconds, ragged last tiles included."""
    monkeypatch.setenv("WX_FF_MIN_WGS", "0")
    monkeypatch.setenv("WX_ATTN_BLOCK", "0")   # (on these small maps the one-launch attention block would take the out-projection and to_qkv)
    cfg = named_config(name)
    sd = synth_state_dict(cfg)
    eng = WXEngine(cfg, "bf16", 0)
    eng.load_state_dict(sd)
    eng.finalize()
    eng.profile(1)
    x = synth_input(cfg)
    y = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    names = {r["name"] for r in eng.profile_read()}
    assert "out_ff_qkv_fused" in names and "out_ff_fused" in names
    check(y, O.forward(cfg, sd, x).numpy(), "bf16")
    monkeypatch.delenv("WX_FF_MIN_WGS")
    monkeypatch.delenv("WX_ATTN_BLOCK")
    eng2 = WXEngine(cfg, "bf16", 0)
    eng2.load_state_dict(sd)
    eng2.finalize()
    eng2.profile(1)
    eng2.forward(torch.from_numpy(x).cuda())
    # small maps: the plain fused block where it yields >= 40 workgroups at C = 128 (stage 1 of the 1-degree grid) and its hidden-split form
    # (+ the split-K finish kernel) on C = 128 / 256 maps of <= 32 pixel tiles -- both win on launch count; the out-projection variants never
    # run there (the attention block kernel owns to_out)
    fused = {r["name"] for r in eng2.profile_read() if "fused" in r["name"]}
    assert fused == ({"ff_fused", "ff_fused_split"} if name == "C1" else {"ff_fused_split"})


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
@pytest.mark.parametrize("name", ["C3S", "C3"])
def test_full_size_vs_reference_golden(name, prec):
    """BASELINE configs at 721x1440 against strided samples + per-channel sums of the real reference."""
    cfg = named_config(name)
    y = get_engine(name, prec).forward(torch.from_numpy(synth_input(cfg)).cuda()).cpu()
    g = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    s = int(g["stride"])
    check(y[0, :, 0, ::s, ::s].numpy(), g["y"], prec)
    a = y[0, :, 0].double()
    npix = a.shape[1] * a.shape[2]
    f32 = prec in ("fp32", "fp32s")
    tol = (2e-5 if f32 else 3e-3) * npix
    np.testing.assert_allclose(a.sum(dim=(1, 2)).numpy(), g["ch_sum"], rtol=0, atol=tol)
    np.testing.assert_allclose((a * a).sum(dim=(1, 2)).numpy(), g["ch_sumsq"], rtol=1e-4 if f32 else 3e-2)
    _engines.pop((name, prec), None)  # free HBM


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
def test_rollout_glue_vs_reference_golden(prec):
    """wx_step x3: forward + TracerFixer + y*std+mean + update_x against the reference's own pieces."""
    g = np.load(os.path.join(GOLD, "rollout_T0.npz"))
    cfg = named_config("T0")
    eng = get_engine("T0", prec)
    mean, std = synth_denorm(cfg.base_output_channels)
    n_prog = cfg.channels * cfg.levels + cfg.surface_channels
    eng.set_denorm(mean, std)
    eng.set_layout(n_prog, int(g["n_static"]), int(g["n_dyn"]))
    eng.set_tracer_fixer(g["tracer_inds"], g["tracer_thres"], None, denorm=False)
    x = torch.from_numpy(synth_input(cfg)).cuda()
    x0 = x.clone()
    for t in (1, 2, 3):
        frc = torch.from_numpy(synth_forcing(cfg, int(g["n_dyn"]), t)).cuda()
        y, yp, xn = eng.step(x, frc)
        if t == 1:
            assert torch.equal(x, x0), "input must not be modified (caller reuses x, rollout_to_netcdf.py:310)"
        if prec in ("fp32", "fp32s"):
            tol = 1e-4 * t
            assert np.abs(y[0, :, 0].cpu().numpy() - g[f"y{t}"]).max() <= tol * np.abs(g[f"y{t}"]).max()
            assert np.abs(yp[0].cpu().numpy() - g[f"yphys{t}"]).max() <= tol * np.abs(g[f"yphys{t}"]).max()
            assert np.abs(xn[0, :, 0].cpu().numpy() - g[f"x{t}"]).max() <= tol * np.abs(g[f"x{t}"]).max()
        else:
            check(y[0, :, 0].cpu().numpy(), g[f"y{t}"], prec) if t == 1 else None
        # exact structural properties in both precisions
        q = y[0, [int(i) for i in g["tracer_inds"]], 0]
        assert float(q.min()) >= float(g["tracer_thres"][0])                      # clamp really applied
        assert torch.equal(xn[0, :n_prog, 0], y[0, :n_prog, 0])                    # prognostic <- y
        assert torch.equal(xn[0, n_prog:n_prog + 2], x[0, n_prog:n_prog + 2])      # static carried
        assert torch.equal(xn[0, n_prog + 2:], frc[0])                             # forcing replaced
        np.testing.assert_allclose(yp[0].cpu().numpy(), y[0, :, 0].cpu().numpy() * std[:, None, None] + mean[:, None, None],
                                   rtol=1e-6, atol=1e-6)
        x = xn
    eng.set_tracer_fixer([], [], None, denorm=False)


def test_tracer_fixer_denorm_matches_oracle():
    cfg = named_config("T0")
    eng = get_engine("T0", "fp32")
    mean, std = synth_denorm(cfg.base_output_channels)
    eng.set_denorm(mean, std)
    inds = list(range(9, 12))
    eng.set_tracer_fixer(inds, [0.2] * 3, [1.5] * 3, denorm=True)
    x = synth_input(cfg)
    y = eng.forward(torch.from_numpy(x).cuda()).cpu()
    ref = O.tracer_fix(O.forward(cfg, synth_state_dict(cfg), x), inds, [0.2] * 3, torch.from_numpy(mean),
                       torch.from_numpy(std), thres_max=[1.5] * 3)
    assert float((y - ref).abs().max()) <= 1e-5
    phys = y[0, inds, 0] * torch.from_numpy(std[inds])[:, None, None] + torch.from_numpy(mean[inds])[:, None, None]
    assert float(phys.min()) >= 0.2 - 1e-5 and float(phys.max()) <= 1.5 + 1e-5
    eng.set_tracer_fixer([], [], None, denorm=False)


def test_properties_full_size_bf16():
    """Size-independent properties at 721x1440 (C3S): determinism, batch consistency, lon-shift equivariance."""
    cfg = named_config("C3S")
    eng = get_engine("C3S", "bf16")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    y1 = eng.forward(x).clone()
    y2 = eng.forward(x)
    assert torch.equal(y1, y2), "two runs on the same input must be bit-identical"
    # the model is NOT lon-shift equivariant in general (windows), but a shift by a whole padded-grid period is identity
    xs = torch.roll(x, shifts=cfg.image_width, dims=-1)
    assert torch.equal(eng.forward(xs), y1)
    # batch of 2 == two singles
    xb = torch.cat([x, torch.flip(x, dims=[1])], dim=0).contiguous()
    yb = eng.forward(xb)
    assert torch.equal(yb[0:1], y1)
    assert torch.equal(yb[1:2], eng.forward(xb[1:2].contiguous()))
    _engines.pop(("C3S", "bf16"), None)


def test_error_behaviour():
    cfg = named_config("T0")
    eng = WXEngine(cfg, "fp32", 0)
    x = torch.from_numpy(synth_input(cfg)).cuda()
    with pytest.raises(WXEngineError, match="not finalized"):
        eng.forward(x)
    sd = synth_state_dict(cfg)
    partial = dict(sd)
    del partial["up_block4.bias"]
    eng.load_state_dict(partial)
    with pytest.raises(WXEngineError, match="up_block4.bias"):
        eng.finalize()
    with pytest.raises(WXEngineError):
        eng.load_state_dict({"up_block4.bias": np.zeros(5, dtype=np.float32)})  # wrong size
    eng.load_state_dict(sd)
    eng.finalize()
    with pytest.raises(WXEngineError):
        eng.forward(x[:, :-1].contiguous())  # wrong channel count
    with pytest.raises(WXEngineError):
        eng.forward(x.cpu())
    with pytest.raises(WXEngineError, match="wx_set_layout"):
        eng.step(x, None)
    # the engine asks for every reference key it uses; the patch-1 cube embedding is never on the path
    assert set(eng.expected_tensors()) == {k for k in cfg.state_spec() if not k.startswith("cube_embedding.")}


def test_model_shim_forward_matches_engine():
    from wxengine.model import WXFormerHIP
    cfg = named_config("T0")
    mc = dict(image_height=37, image_width=72, frames=1, channels=4, surface_channels=4, input_only_channels=4,
              output_only_channels=3, levels=3, dim=[32, 64, 128, 256], depth=[1, 1, 2, 1],
              global_window_size=[4, 2, 2, 1], local_window_size=3,
              cross_embed_kernel_sizes=[[4, 8, 16, 32], [2, 4], [2, 4], [2, 4]], cross_embed_strides=[2, 2, 2, 2],
              padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]),
              post_conf=dict(activate=False))
    m = WXFormerHIP(precision="fp32", **mc).to("cuda").eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg).items()})
    x = torch.from_numpy(synth_input(cfg)).cuda()
    with torch.no_grad():
        y = m(x)
        y4 = m(x[:, :, 0])  # 4-D input tolerated when frames == 1 (benchmark_parallelism.py:53)
    assert y.shape == (1, 19, 1, 37, 72) and y.device == x.device
    assert torch.equal(y, y4)
    check(y.cpu().numpy(), O.forward(cfg, synth_state_dict(cfg), synth_input(cfg)).numpy(), "fp32")
