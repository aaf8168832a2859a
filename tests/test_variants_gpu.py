"""The launch-shape variants the engine picks by itself (split-K, the CrossEmbed chunk split, the merged ConvTranspose parity launch) against
the same engine with the variant switched off: same model, same input, two engines in one process.

* merged parity convs: every output element is the same K walk in the same order -> bit-identical;
* split-K / chunk split: fp32 partial sums are added in a different (fixed) order -> equal to fp32 rounding in the fp32 engine, inside
  bf16 rounding in the bf16 engine; each variant is deterministic (two runs bit-identical).
C1 (BASELINE config 2, the 1-degree grid) is the configuration where all three trigger.
"""
import os

import numpy as np
import pytest
import torch

from wxengine.config import named_config
from wxengine.engine import WXEngine
from wxengine.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu


def _engine(name, prec, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cfg = named_config(name)
        eng = WXEngine(cfg, prec, 0)   # the switches are read when the engine is created
        eng.load_state_dict(synth_state_dict(cfg))
        eng.finalize()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return eng


def _forward(eng, x):
    return eng.forward(x).clone()


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
def test_split_variants_match_unsplit(prec):
    cfg = named_config("C1")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    on = _engine("C1", prec, {})
    off = _engine("C1", prec, {"WX_NO_SPLIT_K": "1", "WX_NO_EMBED_SPLIT": "1"})
    y_on, y_on2, y_off = _forward(on, x), _forward(on, x), _forward(off, x)
    assert torch.equal(y_on, y_on2), "the split launches must be deterministic (fixed summation order)"
    scale = float(y_off.abs().max())
    err = float((y_on - y_off).abs().max())
    if prec in ("fp32", "fp32s"):
        assert err <= 2e-5 * scale, f"split vs unsplit ({prec}): {err:.3e} of {scale:.3e}"
    else:
        l2 = float(torch.linalg.norm((y_on - y_off).double()) / torch.linalg.norm(y_off.double()))
        assert l2 <= 1e-2, f"split vs unsplit (bf16) rel-L2 {l2:.3e}"
    assert not torch.equal(y_on, y_off) or prec in ("fp32", "fp32s"), "the variants should actually differ in summation order (is the split taken?)"


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
def test_merged_parity_convs_bit_identical(prec):
    cfg = named_config("C1")   # type: crossformer -> ConvTranspose k4 s2 p1 as the last up-block
    x = torch.from_numpy(synth_input(cfg)).cuda()
    merged = _engine("C1", prec, {})
    separate = _engine("C1", prec, {"WX_NO_MERGE_PARITY": "1"})
    assert torch.equal(_forward(merged, x), _forward(separate, x))


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
def test_launch_count_reducers_on_small_maps(prec):
    """1-degree grid (launch-bound: ~5 us of dispatch floor per kernel): (a) one convolution per CrossEmbed of stages 1-3, the k = 2
    branch zero-padded into the k = 4 window; (b) the stage-0 k = 4 branch in the spare accumulator rows of the LDS-patch kernel;
    (c) the GroupNorm partial fold inside the apply kernel.  Each against the engine with the reducer off: the added products are
    exact zeros and the folds keep a fixed order, so only the grouping of fp32 sums changes."""
    cfg = named_config("C1")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    on = _engine("C1", prec, {})
    off = _engine("C1", prec, {"WX_NO_EMBED_MERGE": "1", "WX_NO_EMBED_RIDE4": "1", "WX_GN_FOLD_TILES": "0"})
    on.profile(2)
    off.profile(2)
    y_on, y_on2, y_off = _forward(on, x), _forward(on, x), _forward(off, x)
    assert torch.equal(y_on, y_on2)
    scale = float(y_off.abs().max())
    if prec in ("fp32", "fp32s"):
        err = float((y_on - y_off).abs().max())
        assert err <= 2e-5 * scale, f"reducers on vs off ({prec}): {err:.3e} of {scale:.3e}"
    else:
        l2 = float(torch.linalg.norm((y_on - y_off).double()) / torch.linalg.norm(y_off.double()))
        assert l2 <= 1e-2, f"reducers on vs off (bf16) rel-L2 {l2:.3e}"
    c_on = {r["name"]: r["launches"] for r in on.profile_read()}
    c_off = {r["name"]: r["launches"] for r in off.profile_read()}
    for s in (1, 2, 3):   # (two forwards were profiled on `on`, one on `off`)
        assert c_on[f"gemm_embed.s{s}"] == 2 * 1 and c_off[f"gemm_embed.s{s}"] == 2
    assert "gemm_embed.s0" not in c_on and c_off.get("gemm_embed.s0", 0) == 1
    assert sum(v for k, v in c_on.items() if k.startswith("gn_stats")) < 2 * sum(v for k, v in c_off.items() if k.startswith("gn_stats"))


@pytest.mark.parametrize("name", ["T5", "C1"])
def test_persistent_gemm_forced_on_small_maps(name):
    """`gemm_stream_kernel` (wx_gemm_stream.h) only takes C >= 512 layers with >= 4096 rows by itself, i.e. only the 0.25-degree model.
    WX_STREAM_MIN_ROWS=0 forces it onto small maps: T5 (C = 512 on 200 rows, C = 1024 on 50 rows: ragged against the 128- and 160-row
    tiles, a sink-row tile, LN fold / LN fold + GELU into the k-blocked hidden / residual + row partials from the k-blocked hidden) and C1's
    stage 3 (C = 512 on 360 rows).  Checked (a) against the same engine on the 128 x 128 kernel (`WX_NO_STREAM=1`): every GEMM output is the
    same K walk -> the block outputs agree to the bf16 rounding of a differently ordered LayerNorm partial sum at most, (b) per block against
    the fp64-free CPU oracle with the suite's bf16 gate, (c) run-to-run bit-identical (race screen)."""
    from oracle import wxformer_oracle as O
    cfg = named_config(name)
    sd = synth_state_dict(cfg)
    xin = synth_input(cfg)
    x = torch.from_numpy(xin).cuda()
    stream = _engine(name, "bf16", {"WX_STREAM_MIN_ROWS": "0"})
    plain = _engine(name, "bf16", {"WX_NO_STREAM": "1"})
    cap = {}
    y_ref = O.forward(cfg, sd, xin, capture=cap)
    stream.set_debug(True)
    y_s = stream.forward(x).clone()
    got_s = {k: stream.debug_read(k) for k in cap}
    stream.set_debug(False)
    plain.set_debug(True)
    y_p = plain.forward(x).clone()
    got_p = {k: plain.debug_read(k) for k in cap}
    plain.set_debug(False)
    assert torch.equal(y_s, stream.forward(x)), "persistent GEMM: two runs differ (race)"
    for eng, want in ((stream, True), (plain, False)):   # prove which kernel family each engine ran
        eng.profile(3)
        eng.profile_reset()
        eng.forward(x)
        torch.cuda.synchronize()
        tagged = [r["name"] for r in eng.profile_read() if r["name"].endswith(("@stream", "@stream_lc"))]   # _lc: its loader / consumer form
        eng.profile(0)
        assert bool(tagged) == want, tagged
        if want:
            assert {t.split(".")[0] for t in tagged} >= {"gemm_qkv", "gemm_out", "gemm_ff1", "gemm_ff2"}, tagged
    deep = [k for k in cap if k.startswith("layers.2.1.") or k.startswith("layers.3.1.")]
    assert len(deep) >= 6, deep
    n_diff = 0
    for k in deep:
        ref = cap[k][0].numpy().astype(np.float64)
        a, b = got_s[k].astype(np.float64), got_p[k].astype(np.float64)
        l2_ref = np.linalg.norm(a - ref) / np.linalg.norm(ref)
        l2_pl = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert l2_ref <= 2e-2, f"{name} {k}: persistent GEMM vs oracle rel-L2 {l2_ref:.3e}"
        # two bf16 runs whose LayerNorm partial sums are added in a different order differ in single ulps of a few rows, and the
        # blocks downstream decorrelate at bf16 noise level (4-6e-3 rel-L2, the same figure the lat-band tests see)
        assert l2_pl <= 8e-3, f"{name} {k}: persistent vs 128x128 kernel rel-L2 {l2_pl:.3e}"
        n_diff += int(not np.array_equal(a, b))
    l2 = float(torch.linalg.norm((y_s - y_p).double()) / torch.linalg.norm(y_p.double()))
    assert l2 <= 1e-2, f"{name}: forward, persistent vs 128x128 kernel rel-L2 {l2:.3e}"
    yr = y_ref.numpy().astype(np.float64)
    l2o = np.linalg.norm(y_s.cpu().numpy().astype(np.float64) - yr) / np.linalg.norm(yr)
    assert l2o <= 2e-2, f"{name}: forward vs oracle rel-L2 {l2o:.3e}"
    print(f"[stream parity] {name}: {len(deep)} deep-stage captures, {n_diff} differ bitwise from the 128x128 path; y vs plain {l2:.2e}, vs oracle {l2o:.2e}")


@pytest.mark.parametrize("name", ["T5", "C1"])
def test_eight_phase_conv_forced_on_small_maps(name):
    """`gemm8p_kernel` (wx_gemm8p.h, conv form: eight waves / eight phases, 160 x 256 tiles, tap masks + buffer-descriptor LDS-DMA) takes the
    decoder's 3x3 convs with >= 256 output channels on maps of >= 16384 pixels by itself -- the first two UpBlocks of the 0.25-degree model.
    WX_GEMM8P_MIN_ROWS=1 forces it onto T5 (512 -> 512 on 10 x 20 pixels: two ragged 160-row tiles x two N-tiles, K = 4608; 256 -> 256 on
    20 x 40) and C1 (256 -> 256 on 30 x 48): map borders on every side of a tile, rows beyond M in the last tile, GroupNorm partials per
    80-row half tile.  The same switch also routes the decoder's ConvTranspose k2 s2 layers (1x1 form, 2 x 2 pixel scatter in the
    epilogue) through the kernel.  Checked (a) against the same engine on the 128 x 128 kernel (`WX_NO_GEMM8P=1`), (b) per UpBlock against the CPU oracle
    with the suite's bf16 gate, (c) run-to-run bit-identical (race screen), (d) that the kernel is what ran."""
    from oracle import wxformer_oracle as O
    cfg = named_config(name)
    sd = synth_state_dict(cfg)
    xin = synth_input(cfg)
    x = torch.from_numpy(xin).cuda()
    e8 = _engine(name, "bf16", {"WX_GEMM8P_MIN_ROWS": "1"})
    plain = _engine(name, "bf16", {"WX_NO_GEMM8P": "1"})
    cap = {}
    y_ref = O.forward(cfg, sd, xin, capture=cap)
    got = {}
    for eng, key in ((e8, "8p"), (plain, "plain")):
        eng.set_debug(True)
        y = eng.forward(x).clone()
        got[key] = (y, {k: eng.debug_read(k) for k in cap if k.startswith("up_block")})
        eng.set_debug(False)
    assert torch.equal(e8.forward(x), e8.forward(x)), "eight-phase conv: two runs differ (race)"
    n8 = e8.query("gemm8p_launches")
    assert n8 >= 2 and plain.forward(x) is not None and plain.query("gemm8p_launches") == 0, (n8, plain.query("gemm8p_launches"))
    e8.profile(3)
    e8.profile_reset()
    e8.forward(x)
    torch.cuda.synchronize()
    tagged = [r["name"] for r in e8.profile_read() if r["name"].endswith("@gemm8p")]
    e8.profile(0)
    assert tagged and all(t.startswith(("gemm_conv3", "gemm_convT2")) for t in tagged), tagged
    assert any(t.startswith("gemm_conv3") for t in tagged) and any(t.startswith("gemm_convT2") for t in tagged), tagged
    ups = sorted(got["8p"][1])
    assert len(ups) >= 3, ups
    for k in ups:
        ref = cap[k][0].numpy().astype(np.float64)
        a, b = got["8p"][1][k].astype(np.float64), got["plain"][1][k].astype(np.float64)
        l2_ref = np.linalg.norm(a - ref) / np.linalg.norm(ref)
        l2_pl = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert l2_ref <= 2e-2, f"{name} {k}: eight-phase conv vs oracle rel-L2 {l2_ref:.3e}"
        assert l2_pl <= 8e-3, f"{name} {k}: eight-phase conv vs 128x128 kernel rel-L2 {l2_pl:.3e}"
    y8, yp = got["8p"][0], got["plain"][0]
    l2 = float(torch.linalg.norm((y8 - yp).double()) / torch.linalg.norm(yp.double()))
    yr = y_ref.numpy().astype(np.float64)
    l2o = np.linalg.norm(y8.cpu().numpy().astype(np.float64) - yr) / np.linalg.norm(yr)
    assert l2 <= 1e-2 and l2o <= 2e-2, (l2, l2o)
    print(f"[gemm8p conv parity] {name}: {n8} launches; y vs 128x128 path {l2:.2e}, vs oracle {l2o:.2e}")


@pytest.mark.parametrize("name,env", [("T5", {"WX_STREAM_MIN_ROWS": "0"}), ("C3", {})])
def test_attention_sub_block_on_k_blocked_layouts_bit_identical(name, env):
    """Round 6: at C >= 512 on the persistent GEMMs, to_qkv writes q|k|v k-blocked ([C/32][tokens][32] = [head][token][32] at dim_head 32),
    the window attention reads it and writes its output the same way, to_out reads that as its k-blocked operand -- only addresses change,
    so the forward must be BITWISE the row-major chain's (`WX_NO_ATTN_BLK=1`).  T5 forces the persistent GEMMs onto 200- and 50-token maps
    (short 5 x 5 and long 2 x 2 / 1 x 1 windows, ragged GEMM tiles); C3 is the headline model (stage 2: 10 x 10 short and packed 2 x 2 long
    windows on 20 000 tokens; stage 3: 5 000)."""
    cfg = named_config(name)
    x = torch.from_numpy(synth_input(cfg)).cuda()
    blk = _engine(name, "bf16", dict(env))
    row = _engine(name, "bf16", dict(env, WX_NO_ATTN_BLK="1"))
    y_b, y_r = _forward(blk, x), _forward(row, x)
    nb, nr = blk.query("attn_blk"), row.query("attn_blk")
    assert nb >= 2 and nr == 0, (nb, nr)
    assert torch.equal(y_b, y_r), f"{name}: k-blocked attention chain differs from the row-major one (max {float((y_b - y_r).abs().max()):.3e})"
    assert torch.equal(y_b, _forward(blk, x))
    print(f"[attention k-blocked] {name}: {nb} sub-blocks on the blocked layouts, forward bit-identical to the row-major chain")


@pytest.mark.parametrize("name", ["T5", "C1"])
def test_weight_stationary_gemm_forced_on_small_maps(name):
    """`gemm_wreg_kernel` (wx_gemm_wreg.h: weights in registers, activations streamed in 32-row tiles) takes the K = 512 layers on maps
    of 1024 .. 4095 rows by itself -- a lat-band rank's share of the 0.25-degree stage 2.  WX_WREG_MIN_ROWS=0 forces it onto T5's
    stage 2 (C = 512 on 200 rows: ragged last tile, fewer tiles than workgroups) and C1's stage 3 (C = 512 on 360 rows), where its
    three epilogues run: LN fold from partials AND from final row statistics, LN fold + GELU, residual + row partials (16 slots).
    Checked like the persistent kernel: against the engine without it, per block against the oracle, run-to-run, and by family tag."""
    from oracle import wxformer_oracle as O
    cfg = named_config(name)
    sd = synth_state_dict(cfg)
    xin = synth_input(cfg)
    x = torch.from_numpy(xin).cuda()
    wreg = _engine(name, "bf16", {"WX_WREG_MIN_ROWS": "0", "WX_SKINNY_MAX": "0"})   # (the skinny split-K rule would take C1's 360-row layers first)
    plain = _engine(name, "bf16", {"WX_NO_WREG": "1", "WX_SKINNY_MAX": "0"})
    cap = {}
    y_ref = O.forward(cfg, sd, xin, capture=cap)
    outs = {}
    for tag, eng in (("wreg", wreg), ("plain", plain)):
        eng.set_debug(True)
        y = eng.forward(x).clone()
        outs[tag] = (y, {k: eng.debug_read(k) for k in cap})
        eng.set_debug(False)
    assert torch.equal(outs["wreg"][0], wreg.forward(x)), "weight-stationary GEMM: two runs differ (race)"
    for eng, want in ((wreg, True), (plain, False)):
        eng.profile(3)
        eng.profile_reset()
        eng.forward(x)
        torch.cuda.synchronize()
        tagged = [r["name"] for r in eng.profile_read() if r["name"].endswith("@wreg")]
        eng.profile(0)
        assert bool(tagged) == want, tagged
        if want:
            assert {t.split(".")[0] for t in tagged} >= {"gemm_qkv", "gemm_out", "gemm_ff1"}, tagged
    stage = "layers.2.1." if name == "T5" else "layers.3.1."
    deep = [k for k in cap if k.startswith(stage)]
    assert len(deep) >= 4, deep
    for k in deep:
        ref = cap[k][0].numpy().astype(np.float64)
        a, b = outs["wreg"][1][k].astype(np.float64), outs["plain"][1][k].astype(np.float64)
        assert np.linalg.norm(a - ref) / np.linalg.norm(ref) <= 2e-2, k
        assert np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30) <= 8e-3, k
    yr = y_ref.numpy().astype(np.float64)
    l2o = np.linalg.norm(outs["wreg"][0].cpu().numpy().astype(np.float64) - yr) / np.linalg.norm(yr)
    assert l2o <= 2e-2, f"{name}: forward vs oracle rel-L2 {l2o:.3e}"


@pytest.mark.parametrize("name", ["T5", "C1"])
def test_loader_consumer_gemm_forced_on_small_maps(name):
    """The loader / consumer form of `gemm_stream_kernel` (wx_gemm_stream.h, LC: four MFMA waves + four staging waves, 8-stage ring)
    takes the residual layers that are down to one 160 x 128 tile per CU with K >= 1024 -- stage 3 of the 0.25-degree model by itself.
    WX_STREAM_MIN_ROWS=0 brings the persistent kernel onto T5's stage 2 (C = 512 on 200 rows: a ragged last tile) and C1's stage 3
    (360 rows), where FeedForward layer 2 (K = 2048) then runs in that form.  Same arithmetic in the same order as the 4-wave form:
    the forward must be BITWISE equal to the engine with WX_NO_STREAM_LC=1, run-to-run stable, inside the bf16 gate against the
    oracle, and the profile must say which form ran."""
    from oracle import wxformer_oracle as O
    cfg = named_config(name)
    xin = synth_input(cfg)
    x = torch.from_numpy(xin).cuda()
    common = {"WX_STREAM_MIN_ROWS": "0", "WX_SKINNY_MAX": "0", "WX_NO_WREG": "1"}
    lc = _engine(name, "bf16", dict(common))
    plain = _engine(name, "bf16", dict(common, WX_NO_STREAM_LC="1"))
    y_lc, y_plain = lc.forward(x).clone(), plain.forward(x).clone()
    assert torch.equal(y_lc, y_plain), "loader / consumer GEMM: output differs from the 4-wave form"
    assert torch.equal(y_lc, lc.forward(x)), "loader / consumer GEMM: two runs differ (race)"
    for eng, want in ((lc, True), (plain, False)):
        eng.profile(3)
        eng.profile_reset()
        eng.forward(x)
        torch.cuda.synchronize()
        tagged = [r["name"] for r in eng.profile_read() if r["name"].endswith("@stream_lc")]
        eng.profile(0)
        assert bool(tagged) == want, tagged
        if want:
            assert {t.split(".")[0] for t in tagged} >= {"gemm_ff2"}, tagged   # (+ to_out where a stage is 1024 wide: K = 1024)
    yr = O.forward(cfg, synth_state_dict(cfg), xin).numpy().astype(np.float64)
    l2o = np.linalg.norm(y_lc.cpu().numpy().astype(np.float64) - yr) / np.linalg.norm(yr)
    assert l2o <= 2e-2, f"{name}: forward vs oracle rel-L2 {l2o:.3e}"


@pytest.mark.parametrize("name", ["T1", "T5", "C1", "RT"])
def test_attention_block_kernel_opt_in(name):
    """`attn_block_kernel` (wx_attn_block.h: LayerNorm + to_qkv + window attention + to_out + residual in one launch, q|k|v never in
    memory) runs by default only where it measured faster (stage 0 of the 0.25-degree model and launch-bound maps of <= 32768 tokens,
    WX_ATTN_BLOCK=2); WX_ATTN_BLOCK=1 forces it wherever it exists, =0 turns it off.  Its parity is pinned here on every window shape the small configs offer: T1 / T5 5 x 5 windows, short and long (dilated), C = 128
    and 256; C1 3 x 3 short and 4 x 4 long at C = 128 / 256; RT 4 x 4 at C = 128 / 256 -- per block against the CPU oracle (bf16 gate)
    and against the engine with it off; the profile proves which path ran.  (C3's 10 x 10 windows: test_attention_block_full_size.)"""
    from oracle import wxformer_oracle as O
    cfg = named_config(name)
    sd = synth_state_dict(cfg)
    xin = synth_input(cfg)
    x = torch.from_numpy(xin).cuda()
    blk = _engine(name, "bf16", {"WX_ATTN_BLOCK": "1"})
    ref = _engine(name, "bf16", {"WX_ATTN_BLOCK": "0"})
    cap = {}
    y_ref = O.forward(cfg, sd, xin, capture=cap).numpy().astype(np.float64)
    blk.set_debug(True)
    y_b = blk.forward(x).clone()
    got = {k: blk.debug_read(k) for k in cap}
    blk.set_debug(False)
    assert torch.equal(y_b, blk.forward(x)), "attention block: two runs differ (race)"
    for eng, want in ((blk, True), (ref, False)):
        eng.profile(2)
        eng.profile_reset()
        eng.forward(x)
        torch.cuda.synchronize()
        names = [r["name"] for r in eng.profile_read()]
        eng.profile(0)
        assert any(n.startswith("attn_block") for n in names) == want, names
    n_blocks = 0
    for k, v in cap.items():
        if ".1.layers." not in k:
            continue
        r = v[0].numpy().astype(np.float64)
        l2 = np.linalg.norm(got[k].astype(np.float64) - r) / np.linalg.norm(r)
        assert l2 <= 2e-2, f"{name} {k}: attention block vs oracle rel-L2 {l2:.3e}"
        n_blocks += 1
    assert n_blocks >= 8
    yb = y_b.cpu().numpy().astype(np.float64)
    l2 = np.linalg.norm(yb - y_ref) / np.linalg.norm(y_ref)
    assert l2 <= 2e-2 and np.abs(yb - y_ref).max() <= 5e-2 * np.abs(y_ref).max(), f"{name}: rel-L2 {l2:.3e}"
    y_d = ref.forward(x).cpu().numpy().astype(np.float64)
    l2d = np.linalg.norm(yb - y_d) / np.linalg.norm(y_d)
    assert l2d <= 1.5e-2, f"{name}: block path vs default path rel-L2 {l2d:.3e}"


def test_attention_block_full_size():
    """The opt-in attention block at BASELINE config 3's own size (10 x 10 windows = 112-token tiles, 3200 workgroups at C = 128, the
    5 x 5 dilated windows of stage 1 at C = 256): the reference golden's strided samples under the suite's bf16 gate."""
    cfg = named_config("C3")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    eng = _engine("C3", "bf16", {"WX_ATTN_BLOCK": "1"})
    eng.profile(2)
    y = eng.forward(x).cpu()
    rows = {r["name"]: r["launches"] for r in eng.profile_read()}
    assert rows.get("attn_block.s0") == 4 and rows.get("attn_block.s1") == 4, rows
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_C3.npz"))
    s = int(g["stride"])
    got, want = y[0, :, 0, ::s, ::s].numpy().astype(np.float64), g["y"].astype(np.float64)
    l2 = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert l2 <= 2e-2 and np.abs(got - want).max() <= 5e-2 * np.abs(want).max(), f"rel-L2 {l2:.3e}"


@pytest.mark.parametrize("name", ["T5", "C1", "C3"])
def test_two_stream_half_maps_bit_identical(name):
    """Round 5: the sub-block chains of the C >= 512 stages run as two half-maps of whole window rows on two streams (the caller's and an
    engine-owned one, forked / joined by an event pair; wx_engine.hip stage_blocks_two_stream).  Every kernel is row-independent at
    window-row granularity and a half runs the kernels and tiles the whole map would (rule_rows), so the step must be BIT-identical to
    the one-stream schedule (WX_TWO_STREAM=0) -- on the headline model itself (C3: stage 2 = 5 + 5 window rows per short sub-block,
    stage 3 = 3 + 2 window rows forked once) and, forced onto small maps with WX_STREAM_MIN_ROWS=0, on T5 (1 + 1 window rows at stage
    2) and C1's stage 3 -- and bit-identical run to run (race screen: three forwards)."""
    cfg = named_config(name)
    x = torch.from_numpy(synth_input(cfg)).cuda()
    force = {} if name == "C3" else {"WX_STREAM_MIN_ROWS": "0"}
    two = _engine(name, "bf16", {**force, "WX_TWO_STREAM": "1"})
    one = _engine(name, "bf16", {**force, "WX_TWO_STREAM": "0"})
    y1 = _forward(one, x)
    y2 = _forward(two, x)
    assert torch.isfinite(y2).all()
    assert torch.equal(y1, y2), f"{name}: two-stream step differs from the one-stream step (max {float((y1 - y2).abs().max()):.3e})"
    for _ in range(3):
        assert torch.equal(_forward(two, x), y2), f"{name}: two-stream step is not deterministic (race between the streams)"
    # the schedule was actually taken: the engine reports its side stream
    assert two.info().get("two_stream_stages", 0) >= 1 and one.info().get("two_stream_stages", 0) == 0


def test_split_precision_is_what_ran():
    """WX_PREC_FP32_SPLIT ("fp32s"): every implicit GEMM of a forward runs split-bf16 arithmetic (wx_query "split_gemms" counts the
    launches; the exact-f32 engine reports 0), the engine says which precision it was created with, the outputs differ from the exact-f32
    engine's by the split's 2^-17 product error and no more (<= 5e-5 of max|y| on the 1-degree model), and two runs are bit-identical.
    (Lat-band mode takes the precision too: tests/test_latband_gpu.py.)"""
    cfg = named_config("C1")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    sp = _engine("C1", "fp32s", {})
    ex = _engine("C1", "fp32", {})
    ys, ye = _forward(sp, x), _forward(ex, x)
    assert sp.query("precision") == 2 and ex.query("precision") == 0
    assert sp.query("split_gemms") >= 80 and ex.query("split_gemms") == 0
    assert sp.query("launches") > 0
    assert torch.equal(ys, _forward(sp, x)), "split-bf16 forward is not deterministic"
    dev = float((ys - ye).abs().max() / ye.abs().max())
    assert 0.0 < dev <= 5e-5, f"fp32s vs exact-f32 engine: {dev:.3e} of max|y|"


def test_split_feedforward_one_launch():
    """fp32s, C = 128 / 256 stages: the FeedForward sub-block runs as ONE launch (wx_ff_split.h: the hidden tensor never reaches HBM; GELU by
    the Abramowitz-Stegun erf, 4.7e-7 absolute).  Against the two-GEMM form (WX_NO_FF_SPLIT_FUSED=1, libm erff) the forward differs by
    rounding order only (<= 2e-5 of max|y|, inside the mode's 1e-4 budget against the reference), the 64- and 128-token tile forms are
    bit-identical (a token's arithmetic does not depend on its tile), and wx_query says how many sub-blocks took the kernel."""
    cfg = named_config("C1")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    fused = _engine("C1", "fp32s", {})
    pair = _engine("C1", "fp32s", {"WX_NO_FF_SPLIT_FUSED": "1"})
    tw1 = _engine("C1", "fp32s", {"WX_FF_SPLIT_TW": "1"})
    tw2 = _engine("C1", "fp32s", {"WX_FF_SPLIT_TW": "2"})
    yf, yp = _forward(fused, x), _forward(pair, x)
    n_sub = sum(2 * d for d, c in zip(cfg.depth, cfg.dim) if c in (128, 256))
    assert n_sub > 0 and fused.query("ff_split_fused") == n_sub and pair.query("ff_split_fused") == 0
    only128 = _engine("C1", "fp32s", {"WX_NO_FF_SPLIT_256": "1"})
    _forward(only128, x)
    assert only128.query("ff_split_fused") == sum(2 * d for d, c in zip(cfg.depth, cfg.dim) if c == 128)
    assert fused.query("split_gemms") == pair.query("split_gemms")
    assert fused.query("launches") < pair.query("launches")
    dev = float((yf - yp).abs().max() / yp.abs().max())
    assert 0.0 < dev <= 2e-5, f"one-launch FeedForward vs the GEMM pair: {dev:.3e} of max|y|"
    assert torch.equal(_forward(tw1, x), _forward(tw2, x))
    assert torch.equal(yf, _forward(fused, x))


def test_split_feedforward_takes_out_projection():
    """fp32s, C = 128 / 256 stages: the one-launch FeedForward also applies the attention's out-projection + residual in front of it
    (wx_ff_split.h PRE instantiations: x1 = x + Wout . o + bo in the accumulators, LayerNorm statistics of x1 two-pass in registers) --
    no to_out launch, x1 is not written by one kernel and read by the next.  Against the unfused order (WX_NO_FF_SPLIT_PRE=1: to_out as a
    split GEMM with one-pass LayerNorm partials) the forward differs by rounding order only, the GEMM count is the same, the launch count
    drops by one per sub-block, both tile forms agree bit for bit, and two runs are bit-identical."""
    cfg = named_config("C1")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    pre = _engine("C1", "fp32s", {})
    nopre = _engine("C1", "fp32s", {"WX_NO_FF_SPLIT_PRE": "1"})
    yp, yn = _forward(pre, x), _forward(nopre, x)
    n_sub = sum(2 * d for d, c in zip(cfg.depth, cfg.dim) if c in (128, 256))
    assert pre.query("ff_split_pre") == n_sub and nopre.query("ff_split_pre") == 0
    assert pre.query("split_gemms") == nopre.query("split_gemms")
    assert pre.query("launches") <= nopre.query("launches") - n_sub
    assert torch.isfinite(yp).all()
    dev = float((yp - yn).abs().max() / yn.abs().max())
    assert 0.0 < dev <= 2e-5, f"out-projection inside the FeedForward launch vs in front of it: {dev:.3e} of max|y|"
    tw1 = _engine("C1", "fp32s", {"WX_FF_SPLIT_TW": "1"})
    tw2 = _engine("C1", "fp32s", {"WX_FF_SPLIT_TW": "2"})
    assert torch.equal(_forward(tw1, x), _forward(tw2, x))
    assert torch.equal(yp, _forward(pre, x))


def test_split_feedforward_makes_next_qkv():
    """fp32s, C = 128 / 256 stages: the one-launch FeedForward that took the out-projection in front also runs the NEXT attention's
    LayerNorm + to_qkv behind it (wx_ff_split.h POST instantiations: the output rows are still in the accumulators -- statistics two-pass
    in registers, q|k|v as 3C / 128 more layer-1-shaped chunks of the weight ring) -- no to_qkv launch, the rows are not read again.
    Against the unfused order (WX_NO_FF_SPLIT_POST=1) the forward differs by rounding order only, the GEMM count is the same, one launch
    per fused sub-block is gone, both tile forms agree bit for bit, and two runs are bit-identical."""
    cfg = named_config("C1")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    post = _engine("C1", "fp32s", {})
    nopost = _engine("C1", "fp32s", {"WX_NO_FF_SPLIT_POST": "1"})
    yp, yn = _forward(post, x), _forward(nopost, x)
    # every sub-block of a C = 128 / 256 stage whose successor in the stage is an attention with windows of more than one token
    n_post = 0
    for d, c, gw in zip(cfg.depth, cfg.dim, cfg.global_window_size):
        if c in (128, 256):
            n_post += (d if gw > 1 else 0) + (d - 1)
    assert n_post > 0 and post.query("ff_split_post") == n_post and nopost.query("ff_split_post") == 0
    assert post.query("split_gemms") == nopost.query("split_gemms")
    assert post.query("launches") <= nopost.query("launches") - n_post
    assert torch.isfinite(yp).all()
    dev = float((yp - yn).abs().max() / yn.abs().max())
    assert 0.0 < dev <= 2e-5, f"to_qkv behind the FeedForward launch vs its own launch: {dev:.3e} of max|y|"
    tw1 = _engine("C1", "fp32s", {"WX_FF_SPLIT_TW": "1"})
    tw2 = _engine("C1", "fp32s", {"WX_FF_SPLIT_TW": "2"})
    assert torch.equal(_forward(tw1, x), _forward(tw2, x))
    assert torch.equal(yp, _forward(post, x))


@pytest.mark.parametrize("precision", ["bf16", "fp32s"])
def test_embed_branch_on_side_stream_bit_identical(precision):
    """Stage-0 CrossEmbed of the 0.25-degree widths: the k = 4 branch does not fit the patch kernel's accumulator rows and runs as its own
    implicit GEMM -- with WX_EMBED_SIDE=1 on the engine's side stream, beside the patch launch (it reads the packed input only and writes a
    channel range of its own; fork / join = one event pair inside cross_embed; measured a tie, so OFF by default).  Same kernels, same
    launches: the forward must be BIT-identical to the one-stream order and bit-identical run to run (race screen: three forwards)."""
    cfg = named_config("C3S")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    side = _engine("C3S", precision, {"WX_EMBED_SIDE": "1"})
    one = _engine("C3S", precision, {})
    y1 = _forward(one, x)
    y2 = _forward(side, x)
    assert torch.isfinite(y2).all()
    assert torch.equal(y1, y2), f"side-stream CrossEmbed branch differs (max {float((y1 - y2).abs().max()):.3e})"
    for _ in range(3):
        assert torch.equal(_forward(side, x), y2), "side-stream CrossEmbed branch is not deterministic (race between the streams)"
    assert side.query("launches") == one.query("launches")
