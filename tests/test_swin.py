"""SURVEY.md 8(f) row 4 -- the second architecture's hot op: Swin (V2-Cr) shifted-window attention.

CPU: oracle/swin_oracle.py against tests/golden/swin_attention.npz, the output of the reference's own `_make_attention_mask`,
`_shifted_window_attn`, `window_partition / window_reverse` and `WindowMultiHeadAttention.forward` (tools/make_goldens.py::swin_golden;
the position-bias table is an input of the fixture because the reference computes it with timm, which is not vendored).
GPU: the HIP operator (`wx_winattn_*` through wxengine.swin.WindowAttention) against the same goldens:
    fp32 (exact-f32 MFMA): max|out - ref| <= 1e-4 * max|ref|;   bf16: rel-L2 <= 2e-2, max err <= 5e-2 * max|ref|."""
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
