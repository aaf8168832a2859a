"""SURVEY.md 8(f) row 4 -- the second architecture's hot op: Swin (V2-Cr) shifted-window attention.

CPU: oracle/swin_oracle.py against tests/golden/swin_attention.npz, the output of the reference's own `_make_attention_mask`,
`_shifted_window_attn`, `window_partition / window_reverse` and `WindowMultiHeadAttention.forward` (tools/make_goldens.py::swin_golden;
the position-bias table is an input of the fixture because the reference computes it with timm, which is not vendored).
GPU: the HIP operator (`wx_winattn_*` through wxengine.swin.WindowAttention) against the same goldens:
    fp32 (exact-f32 MFMA): max|out - ref| <= 1e-4 * max|ref|;   bf16: rel-L2 <= 2e-2, max err <= 5e-2 * max|ref|."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import swin_oracle as S

GOLD = os.path.join(os.path.dirname(__file__), "golden", "swin_attention.npz")
CASES = ["rect_shift", "fuxi_like", "rect_noshift", "lat_only_shift"]


def load(name):
    g = np.load(GOLD)
    H, W, wy, wx, sy, sx, heads, hd = (int(v) for v in g[f"{name}/geom"])
    t = {k: torch.from_numpy(g[f"{name}/{k}"]) for k in ("x", "qkv_w", "qkv_b", "bias", "logit_scale_raw", "core", "mask")}
    return (H, W), (wy, wx), (sy, sx), heads, hd, t


@pytest.mark.parametrize("name", CASES)
def test_oracle_core_and_mask_match_the_reference(name):
    feat, ws, shift, heads, hd, t = load(name)
    qkv = torch.nn.functional.linear(t["x"], t["qkv_w"], t["qkv_b"])
    out = S.window_attention_core(qkv, heads, ws, shift, t["bias"], S.effective_logit_scale(t["logit_scale_raw"]))
    assert out.shape == t["core"].shape
    assert (out - t["core"]).abs().max() <= 2e-6 * t["core"].abs().max()
    m = S.shift_mask(feat, ws, shift)
    if any(shift):
        assert torch.equal(m, t["mask"])
        assert (m == -100).any() == (shift[0] > 0)       # longitude-only shifts need no mask: the map is periodic there
    else:
        assert m is None and t["mask"].numel() == 0
    # fp64 evaluation of the same oracle: the fixture's fp32 noise floor
    o64 = S.window_attention_core(qkv.double(), heads, ws, shift, t["bias"].double(), S.effective_logit_scale(t["logit_scale_raw"]).double())
    assert (o64 - t["core"].double()).abs().max() <= 5e-6 * t["core"].abs().max()


def test_host_bias_table_matches_the_oracle_restatement():
    from wxengine.swin import effective_logit_scale, relative_position_bias
    gen = torch.Generator().manual_seed(5)
    heads, ws = 3, (4, 6)
    sd = {"meta_mlp.fc1.weight": torch.randn(16, 2, generator=gen), "meta_mlp.fc1.bias": torch.randn(16, generator=gen),
          "meta_mlp.fc2.weight": torch.randn(heads, 16, generator=gen), "meta_mlp.fc2.bias": torch.randn(heads, generator=gen)}
    want = S.relative_position_bias(sd, "", ws, heads, torch.float64).numpy()
    got = relative_position_bias(*(sd[k].numpy() for k in sd), ws)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    raw = torch.tensor([0.5, 2.3, 9.0])
    np.testing.assert_allclose(effective_logit_scale(raw.numpy()), S.effective_logit_scale(raw).numpy(), rtol=1e-6)
    assert effective_logit_scale([9.0])[0] == pytest.approx(100.0)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_hip_window_attention_vs_reference_golden(name, prec):
    from wxengine.swin import WindowAttention, effective_logit_scale
    feat, ws, shift, heads, hd, t = load(name)
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    qkv = torch.nn.functional.linear(t["x"], t["qkv_w"], t["qkv_b"])
    op = WindowAttention(feat, heads, hd, ws, shift, bias=t["bias"].numpy(), logit_scale=effective_logit_scale(t["logit_scale_raw"].numpy()),
                         precision=prec)
    out = op(qkv.to(dt).cuda().contiguous()).float().cpu()
    ref = t["core"]
    if prec == "bf16":      # the operator sees bf16 q | k | v: compare against the oracle on the SAME rounded inputs, then the golden
        ref_r = S.window_attention_core(qkv.to(dt).float(), heads, ws, shift, t["bias"], S.effective_logit_scale(t["logit_scale_raw"]))
        l2 = ((out - ref_r).norm() / ref_r.norm()).item()
        assert l2 <= 1e-2, f"bf16 vs oracle on rounded inputs: rel-L2 {l2:.3e}"
        l2 = ((out - ref).norm() / ref.norm()).item()
        assert l2 <= 2e-2 and (out - ref).abs().max() <= 5e-2 * ref.abs().max(), f"bf16 rel-L2 {l2:.3e}"
    else:
        assert (out - ref).abs().max() <= 1e-4 * ref.abs().max(), f"fp32 max err {(out - ref).abs().max():.3e}"


@pytest.mark.gpu
def test_hip_window_attention_dot_product_mode_and_errors():
    from wxengine.engine import WXEngineError
    from wxengine.swin import WindowAttention
    gen = torch.Generator().manual_seed(3)
    feat, ws, heads, hd = (8, 12), (4, 4), 2, 64
    qkv = torch.randn(feat[0], feat[1], 3 * heads * hd, generator=gen)
    for prec, dt, tol in (("fp32", torch.float32, 1e-4), ("bf16", torch.bfloat16, 3e-2)):
        q = qkv.to(dt)
        op = WindowAttention(feat, heads, hd, ws, (0, 0), bias=None, precision=prec)           # plain softmax(q k^T / sqrt(d)) v
        out = op(q.cuda().contiguous()).float().cpu()
        ref = S.window_attention_core(q.float(), heads, ws, (0, 0), None)
        assert (out - ref).abs().max() <= tol * ref.abs().max()
    with pytest.raises(WXEngineError):
        WindowAttention((8, 12), 2, 48, (4, 4))                 # head_dim not supported
    with pytest.raises(WXEngineError):
        WindowAttention((9, 12), 2, 32, (4, 4))                 # window does not divide the map
    with pytest.raises(WXEngineError):
        op(qkv.cuda())                                         # wrong dtype for a bf16 operator


@pytest.mark.gpu
def test_fuxi_sized_throughput_smoke():
    """BASELINE config 5 shape (fuxi_6h: dim 1024, 8 heads of 128, window 7, 84 x 168 tokens after padding): finite, and a timing line
    for DESIGN.md -- throughput only, FuXi itself needs timm and stays unpinned."""
    from wxengine.swin import WindowAttention
    feat, heads, hd, ws = (84, 168), 8, 128, (7, 7)
    gen = torch.Generator().manual_seed(9)
    qkv = (torch.randn(feat[0], feat[1], 3 * heads * hd, generator=gen) * 0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(heads, 49, 49, generator=gen).numpy() * 0.2
    for shift in ((0, 0), (3, 3)):
        op = WindowAttention(feat, heads, hd, ws, shift, bias=bias, logit_scale=np.full(heads, 10.0, np.float32), precision="bf16")
        out = op(qkv)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            op(qkv, out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50.0
        gb = 4 * feat[0] * feat[1] * heads * hd * 2 / 1e9
        print(f"\\nFuXi-sized window attention shift={shift}: {us:.1f} us per launch, {gb / us * 1e6:.0f} GB/s of q|k|v|out traffic")
        assert torch.isfinite(out.float()).all()


# ---- the whole block and a two-block stage (part 2) ---------------------------------------------------------
BLOCK_GOLD = os.path.join(os.path.dirname(__file__), "golden", "swin_block.npz")
BLOCK_CASES = ["rect", "fuxi_like", "clipped"]


def load_block(name):
    g = np.load(BLOCK_GOLD)
    H, W, wy, wx, heads, hd, depth = (int(v) for v in g[f"{name}/geom"])
    pre = f"{name}/sd/"
    sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    ys = [torch.from_numpy(g[f"{name}/y{i}"]) for i in range(depth)]
    return (H, W), (wy, wx), heads, hd, depth, torch.from_numpy(g[f"{name}/x"]), sd, ys


def _clip(feat, ws):   # swin.py:405-409
    w = tuple(f if f <= v else v for f, v in zip(feat, ws))
    return w


@pytest.mark.parametrize("name", BLOCK_CASES)
def test_oracle_block_matches_the_reference_forward(name):
    """oracle/swin_oracle.py::block against the reference's own SwinTransformerV2CrBlock.forward (two blocks: unshifted, shifted)."""
    feat, ws_t, heads, hd, depth, x, sd, ys = load_block(name)
    ws = _clip(feat, ws_t)
    cur = x
    for i in range(depth):
        shift = tuple(0 if (i % 2 == 0 or f <= w) else w // 2 for f, w in zip(feat, ws))
        cur = S.block(cur, sd, heads, ws, shift, prefix=f"blocks.{i}.")
        assert (cur - ys[i]).abs().max() <= 2e-5 * ys[i].abs().max(), (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
@pytest.mark.parametrize("name", BLOCK_CASES)
def test_hip_swin_stage_matches_the_reference_blocks(name, prec):
    """`wx_swin_*` (wxengine.swin.SwinStage) on the reference block's state dict: the stage output after one block (a depth-1 stage)
    and after two (unshifted + shifted) against the reference forward; fp32 1e-4 * max, bf16 rel-L2 2e-2."""
    from wxengine.swin import SwinStage
    feat, ws_t, heads, hd, depth, x, sd, ys = load_block(name)
    dt = torch.bfloat16 if prec == "bf16" else torch.float32   # fp32s: fp32 storage, split-bf16 GEMM arithmetic, the fp32 gate
    for d in (1, depth):
        st = SwinStage(dim=heads * hd, depth=d, num_heads=heads, feat_size=feat, window_size=ws_t, precision=prec)
        st.load_state_dict(sd)
        xin = x.to(dt).cuda().contiguous()
        keep = xin.clone()
        y = st(xin)
        assert torch.equal(xin, keep), "the input must not be modified when out is a different tensor"
        ref = ys[d - 1].double()
        got = y.float().cpu().double()
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        if prec in ("fp32", "fp32s"):
            assert err <= 1e-4 * scale, f"{name} depth {d}: fp32 max err {err:.3e} of {scale:.3e}"
        else:
            l2 = float((got - ref).norm() / ref.norm())
            assert l2 <= 2e-2 and err <= 6e-2 * scale, f"{name} depth {d}: bf16 rel-L2 {l2:.3e}, max err {err:.3e} of {scale:.3e}"
        y2 = st(xin, out=xin)          # in place
        assert torch.equal(y2, y)


@pytest.mark.gpu
def test_fuxi_sized_stage_properties_and_throughput():
    """BASELINE config 5 (FuXi-6h 0.25 degree, config/gen_1/arXiv_2024/fuxi_6h_single_step.yml): 640 x 1280 / patch 4 / down 2 ->
    80 x 160 tokens zero-padded to 84 x 161 (fuxi.py:231-238), dim 1024, 8 heads of 128, 7 x 7 windows, depth 16.  FuXi's own stage
    is timm's class (not vendored: PARITY UNPINNED); this is a full-size property test of the engine's V2-Cr stage at
    that shape plus a throughput line: (a) finite, deterministic; (b) with every norm gain = 0 the post-norm branches vanish and the
    stage is the identity (bit exact); (c) the first two blocks of the 16-block stage equal a 2-block stage with the same weights."""
    import time
    from wxengine.swin import SwinStage
    feat, dim, heads, ws, depth = (84, 161), 1024, 8, 7, 16
    g = torch.Generator().manual_seed(77)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc  # noqa: E731
    sd = {}
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = r(3 * dim, dim, sc=dim ** -0.5), r(3 * dim, sc=0.1)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = r(dim, dim, sc=dim ** -0.5), r(dim, sc=0.1)
        sd[p + "attn.meta_mlp.fc1.weight"], sd[p + "attn.meta_mlp.fc1.bias"] = r(64, 2, sc=0.7), r(64, sc=0.1)
        sd[p + "attn.meta_mlp.fc2.weight"], sd[p + "attn.meta_mlp.fc2.bias"] = r(heads, 64, sc=0.15), r(heads, sc=0.1)
        sd[p + "attn.logit_scale"] = torch.log(10 * torch.ones(heads))
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = 0.3 + r(dim, sc=0.05), r(dim, sc=0.02)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = r(4 * dim, dim, sc=dim ** -0.5), r(4 * dim, sc=0.1)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = r(dim, 4 * dim, sc=(4 * dim) ** -0.5), r(dim, sc=0.1)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = 0.3 + r(dim, sc=0.05), r(dim, sc=0.02)
    st = SwinStage(dim=dim, depth=depth, num_heads=heads, feat_size=feat, window_size=ws, precision="bf16")
    st.load_state_dict(sd)
    x = r(feat[0], feat[1], dim).to(torch.bfloat16).cuda()
    y = st(x)
    assert torch.isfinite(y.float()).all()
    assert torch.equal(y, st(x)), "two runs differ"
    st2 = SwinStage(dim=dim, depth=2, num_heads=heads, feat_size=feat, window_size=ws, precision="bf16")
    st2.load_state_dict(sd)
    sd0 = dict(sd)
    for i in range(depth):
        for n in ("norm1", "norm2"):
            sd0[f"blocks.{i}.{n}.weight"] = torch.zeros(dim)
            sd0[f"blocks.{i}.{n}.bias"] = torch.zeros(dim)
    y2 = st2(x)
    st2.load_state_dict(sd0)
    assert torch.equal(st2(x), x), "zero norm gains: every residual branch must vanish"
    # the 2-block prefix: run the 16-block stage with blocks 2.. neutralised
    sdp = dict(sd)
    for i in range(2, depth):
        for n in ("norm1", "norm2"):
            sdp[f"blocks.{i}.{n}.weight"] = torch.zeros(dim)
            sdp[f"blocks.{i}.{n}.bias"] = torch.zeros(dim)
    st.load_state_dict(sdp)
    assert torch.equal(st(x), y2)
    st.load_state_dict(sd)
    for _ in range(2):
        st(x, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        st(x, out=y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"\n[fuxi-sized stage] 84x161 tokens, dim 1024, 8 heads, 7x7 windows, depth 16, bf16: {dt * 1e3:.2f} ms per stage pass, "
          f"{st.flops / dt / 1e12:.0f} TFLOP/s algorithmic (parity unpinned: FuXi's stage is timm's class)")


@pytest.mark.gpu
def test_stage_linear_layers_on_the_persistent_gemm(monkeypatch):
    """At FuXi's size (>= 4096 tokens, C >= 512) the stage's four Linear layers run on gemm_stream_kernel (k-blocked weight copies,
    bias / bias + GELU epilogues) instead of the 128 x 128 tile kernel.  WX_SWIN_STREAM_MIN_ROWS=0 forces that path onto a map the CPU
    oracle finishes in seconds (14 x 21 tokens, C = 512, 4 heads of 128, 7 x 7 windows, 3 blocks; 294 rows: ragged against the 128-
    and 160-row tiles): checked against the oracle's block (pinned to the reference by swin_block.npz) under the bf16 gate, and against
    the same stage on the tile kernel (WX_SWIN_NO_STREAM=1)."""
    from wxengine.swin import SwinStage
    feat, dim, heads, ws, depth = (14, 21), 512, 4, 7, 3
    g = torch.Generator().manual_seed(31)
    r = lambda *s_, sc=1.0: torch.randn(*s_, generator=g) * sc  # noqa: E731
    sd = {}
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = r(3 * dim, dim, sc=dim ** -0.5), r(3 * dim, sc=0.1)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = r(dim, dim, sc=dim ** -0.5), r(dim, sc=0.1)
        sd[p + "attn.meta_mlp.fc1.weight"], sd[p + "attn.meta_mlp.fc1.bias"] = r(32, 2, sc=0.7), r(32, sc=0.1)
        sd[p + "attn.meta_mlp.fc2.weight"], sd[p + "attn.meta_mlp.fc2.bias"] = r(heads, 32, sc=0.15), r(heads, sc=0.1)
        sd[p + "attn.logit_scale"] = torch.log(10 * torch.ones(heads)) + r(heads, sc=0.2)
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = 1.0 + r(dim, sc=0.2), r(dim, sc=0.1)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = r(4 * dim, dim, sc=dim ** -0.5), r(4 * dim, sc=0.1)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = r(dim, 4 * dim, sc=(4 * dim) ** -0.5), r(dim, sc=0.1)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = 1.0 + r(dim, sc=0.2), r(dim, sc=0.1)
    x = r(feat[0], feat[1], dim)
    ref = x
    for i in range(depth):
        shift = (0, 0) if i % 2 == 0 else (ws // 2, ws // 2)
        ref = S.block(ref, sd, heads, (ws, ws), shift, prefix=f"blocks.{i}.")
    outs = {}
    for mode, env in (("stream", {"WX_SWIN_STREAM_MIN_ROWS": "0"}), ("tile", {"WX_SWIN_NO_STREAM": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        st = SwinStage(dim=dim, depth=depth, num_heads=heads, feat_size=feat, window_size=ws, precision="bf16")
        for k in env:
            monkeypatch.delenv(k)
        st.load_state_dict(sd)
        y = st(x.to(torch.bfloat16).cuda())
        assert torch.equal(y, st(x.to(torch.bfloat16).cuda())), f"{mode}: two runs differ"
        outs[mode] = y.float().cpu()
        l2 = ((outs[mode] - ref).norm() / ref.norm()).item()
        assert l2 <= 2e-2 and (outs[mode] - ref).abs().max() <= 6e-2 * ref.abs().max(), f"{mode}: bf16 rel-L2 {l2:.3e}"
    l2 = ((outs["stream"] - outs["tile"]).norm() / outs["tile"].norm()).item()
    assert l2 <= 1e-2, f"persistent GEMM vs tile kernel rel-L2 {l2:.3e}"


# ---- Attend (credit/attend.py:94-120): the non-windowed mode ----------------------------------------------------------------------------
ATTEND_GOLD = os.path.join(os.path.dirname(__file__), "golden", "attend.npz")
ATTEND_CASES = ["n64_d32", "n128_d64_scaled", "n100_d32"]


def load_attend(name):
    g = np.load(ATTEND_GOLD)
    sc = float(g[f"{name}/scale"][0])
    return tuple(torch.from_numpy(g[f"{name}/{k}"]) for k in ("q", "k", "v", "out")) + (None if np.isnan(sc) else sc,)


@pytest.mark.parametrize("name", ATTEND_CASES)
def test_oracle_attend_matches_the_reference_class(name):
    q, k, v, ref, scale = load_attend(name)
    out = S.attend(q, k, v, scale)
    assert (out - ref).abs().max() <= 2e-6 * ref.abs().max()
    # the same thing as one window per batch item of the window-attention oracle (the mapping the HIP `Attend` uses)
    from wxengine.swin import Attend
    b, h, n, d = q.shape
    wy, wx = Attend._window(n)
    qkv = torch.stack((q, k, v), 0).permute(1, 3, 0, 2, 4).reshape(b * wy, wx, 3 * h * d)
    o = S.window_attention_core(qkv, h, (wy, wx), (0, 0), None, None, scale if scale is not None else d ** -0.5)
    o = o.reshape(b, n, h, d).permute(0, 2, 1, 3)
    assert (o - ref).abs().max() <= 2e-6 * ref.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ATTEND_CASES)
def test_hip_attend_vs_reference_golden(name, prec):
    from wxengine.engine import WXEngineError
    from wxengine.swin import Attend
    q, k, v, ref, scale = load_attend(name)
    att = Attend(flash=True, scale=scale, precision=prec)
    out = att(q.cuda(), k.cuda(), v.cuda()).cpu()
    assert out.shape == ref.shape and out.dtype == torch.float32
    if prec == "fp32":
        assert (out - ref).abs().max() <= 1e-4 * ref.abs().max(), f"{(out - ref).abs().max():.3e}"
    else:
        l2 = ((out - ref).norm() / ref.norm()).item()
        assert l2 <= 2e-2 and (out - ref).abs().max() <= 5e-2 * ref.abs().max(), f"bf16 rel-L2 {l2:.3e}"
    with pytest.raises(ValueError):
        Attend(dropout=0.1)
    with pytest.raises(WXEngineError):
        att(torch.zeros(1, 2, 200, 32).cuda(), torch.zeros(1, 2, 200, 32).cuda(), torch.zeros(1, 2, 200, 32).cuda())


# ---- timm's Swin V2 block: the block FuXi's stage is REALLY made of (fuxi.py:4-5, 250-260).  timm cannot be installed here, so these
# ---- tests pin the engine to oracle/swin_oracle.py::block_timm (a restatement of timm's published block; header: "parity unpinned"),
# ---- and the restatement to everything that can be checked without timm.
def timm_stage_dict(dim, heads, depth, seed=0):
    """Random stage weights under timm's key names (effective weights: no spectral-norm triples at this level)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale  # noqa: E731
    sd = {}
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "attn.logit_scale"] = math.log(10.0) + 0.3 * r(heads, 1, 1)
        sd[p + "attn.q_bias"], sd[p + "attn.v_bias"] = r(dim, scale=0.2), r(dim, scale=0.2)
        sd[p + "attn.cpb_mlp.0.weight"], sd[p + "attn.cpb_mlp.0.bias"] = r(512, 2, scale=0.7), r(512, scale=0.3)
        sd[p + "attn.cpb_mlp.2.weight"] = r(heads, 512, scale=0.08)
        sd[p + "attn.qkv.weight"] = r(3 * dim, dim, scale=dim ** -0.5)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = r(dim, dim, scale=dim ** -0.5), r(dim, scale=0.1)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = r(4 * dim, dim, scale=dim ** -0.5), r(4 * dim, scale=0.1)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = r(dim, 4 * dim, scale=(4 * dim) ** -0.5), r(dim, scale=0.1)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1.0 + 0.1 * r(dim), 0.1 * r(dim)
    return sd


TIMM_CASES = {   # feat, window, heads, head_dim, depth
    "rect_both_axes": ((12, 24), 4, 2, 32, 2),       # shift (2, 2): latitude AND longitude seams in the last window row / column
    "fuxi_like": ((14, 21), 7, 2, 128, 3),           # FuXi's 7 x 7 windows and 128-wide heads
    "lat_clipped": ((4, 16), 4, 2, 64, 2),           # the map is one window tall: that axis is neither clipped further nor shifted
}


def test_timm_mask_and_bias_restatement():
    """What can be said about timm's mask / bias without timm: (1) with a latitude-only shift the 3 x 3-slice mask IS the reference's
    own V2-Cr mask (swin.py:411-427, pinned by swin_attention.npz); (2) inside any window it has at most 2 x 2 regions, only in the
    last window row / column -- the property the kernel's two limits rely on; (3) the host table (numpy float64) equals the oracle's."""
    for feat, ws, sh in (((12, 24), (4, 4), (2, 0)), ((14, 21), (7, 7), (3, 0))):
        assert torch.equal(S.shift_mask_timm(feat, ws, sh), S.shift_mask(feat, ws, sh))
    feat, ws, sh = (12, 24), (4, 8), (2, 4)
    m = S.shift_mask_timm(feat, ws, sh).reshape(feat[0] // ws[0], feat[1] // ws[1], ws[0] * ws[1], ws[0] * ws[1])
    for wy in range(m.shape[0]):
        for wx in range(m.shape[1]):
            t = torch.arange(ws[0] * ws[1])
            reg = (t // ws[1] >= feat[0] - sh[0] - wy * ws[0]).long() + 2 * (t % ws[1] >= feat[1] - sh[1] - wx * ws[1]).long()
            want = torch.where(reg[:, None] != reg[None, :], -100.0, 0.0)
            assert torch.equal(m[wy, wx], want), (wy, wx)
            if wy < m.shape[0] - 1 and wx < m.shape[1] - 1:
                assert not m[wy, wx].any()
    from wxengine.swin import cpb_position_bias
    sd = timm_stage_dict(64, 2, 1)
    for ws in ((4, 4), (7, 7), (4, 8)):
        host = cpb_position_bias(sd["blocks.0.attn.cpb_mlp.0.weight"].numpy(), sd["blocks.0.attn.cpb_mlp.0.bias"].numpy(),
                                 sd["blocks.0.attn.cpb_mlp.2.weight"].numpy(), ws)
        orc = S.cpb_position_bias({k: v.double() for k, v in sd.items()}, "blocks.0.attn.", ws, 2, torch.float64)
        assert host.shape == (2, ws[0] * ws[1], ws[0] * ws[1]) and 0.0 <= host.min() and host.max() <= 16.0
        np.testing.assert_allclose(host, orc.numpy(), rtol=0, atol=2e-6)
        assert np.allclose(host[:, 0, 0], host[:, 5, 5])        # a function of the offset only: the diagonal is constant


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", sorted(TIMM_CASES))
def test_hip_swin_stage_timm_variant_vs_oracle(name, prec):
    """wxengine.swin.SwinStage(variant="timm") -- q / v bias, 16 sigmoid(cpb) table, both-axes mask -- against block_timm / stage_timm
    after one block and after the whole stage; fp32 1e-4 * max, bf16 rel-L2 2e-2 (the gates of the pinned V2-Cr variant)."""
    from wxengine.swin import SwinStage
    feat, window, heads, hd, depth = TIMM_CASES[name]
    dim = heads * hd
    sd = timm_stage_dict(dim, heads, depth, seed=len(name))
    x = torch.randn(feat[0], feat[1], dim, generator=torch.Generator().manual_seed(3))
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    for d in (1, depth):
        st = SwinStage(dim=dim, depth=d, num_heads=heads, feat_size=feat, window_size=window, precision=prec, variant="timm")
        st.load_state_dict(sd)
        got = st(x.to(dt).cuda().contiguous()).float().cpu().double()
        ref = S.stage_timm(x.double(), {k: v.double() for k, v in sd.items()}, heads, window, d)
        scale, err = float(ref.abs().max()), float((got - ref).abs().max())
        if prec == "fp32":
            assert err <= 1e-4 * scale, f"{name} depth {d}: fp32 max err {err:.3e} of {scale:.3e}"
        else:
            l2 = float((got - ref).norm() / ref.norm())
            assert l2 <= 2e-2 and err <= 6e-2 * scale, f"{name} depth {d}: bf16 rel-L2 {l2:.3e}, max err {err:.3e} of {scale:.3e}"
    # the longitude seam matters: the V2-Cr mask on the same weights gives a different answer wherever a window straddles it
    if name == "rect_both_axes":
        from wxengine.swin import WindowAttention, cpb_position_bias, effective_logit_scale
        qkv = torch.randn(feat[0], feat[1], 3 * dim, generator=torch.Generator().manual_seed(5)).cuda()
        kw = dict(feat=feat, heads=heads, head_dim=hd, window=window, shift=(2, 2), precision="fp32",
                  bias=cpb_position_bias(sd["blocks.1.attn.cpb_mlp.0.weight"].numpy(), sd["blocks.1.attn.cpb_mlp.0.bias"].numpy(),
                                         sd["blocks.1.attn.cpb_mlp.2.weight"].numpy(), (window, window)),
                  logit_scale=effective_logit_scale(sd["blocks.1.attn.logit_scale"].numpy().ravel()))
        both, lat = WindowAttention(mask_axes=3, **kw)(qkv).cpu(), WindowAttention(mask_axes=1, **kw)(qkv).cpu()
        seam = torch.tensor([(c - 2) % feat[1] >= feat[1] - window for c in range(feat[1])])   # original columns inside the last rolled window column
        assert torch.equal(both[:, ~seam], lat[:, ~seam])
        assert (both[:, seam] - lat[:, seam]).abs().max() > 1e-3
