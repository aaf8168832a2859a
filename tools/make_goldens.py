#!/usr/bin/env python
"""Generate golden fixtures from the REAL reference (dev container only).

Imports NCAR/miles-credit from /root/reference (read-only, via tools/oracle_stub.py),
loads the build's synthetic name-keyed weights (wxengine.synth) with strict=True into
`credit.models.crossformer.CrossFormer`, runs it on CPU fp32 and stores outputs as
small .npz files under tests/golden/.  Only data (inputs are regenerated from the
keyed RNG; outputs / strided samples / per-channel statistics) is written — no
reference source.

    python tools/make_goldens.py [--only T0,T1,C1,C3S,C3,pad,glue]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "miles-credit_amd"), ROOT]

import oracle_stub  # noqa: E402

oracle_stub.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from wxengine.config import named_config  # noqa: E402
from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def reference_model(cfg, post_conf=None, family="base"):
    if getattr(cfg, "arch", "crossformer") == "wxformer":
        from credit.models.wxformer.crossformer import CrossFormer
    else:
        from credit.models.crossformer import CrossFormer
    m = CrossFormer(
        image_height=cfg.image_height, image_width=cfg.image_width, frames=cfg.frames, channels=cfg.channels,
        surface_channels=cfg.surface_channels, input_only_channels=cfg.input_only_channels,
        output_only_channels=cfg.output_only_channels, levels=cfg.levels, dim=cfg.dim, depth=cfg.depth,
        dim_head=cfg.dim_head, global_window_size=cfg.global_window_size,
        local_window_size=cfg.local_window_size[0], cross_embed_kernel_sizes=cfg.cross_embed_kernel_sizes,
        cross_embed_strides=cfg.cross_embed_strides, use_spectral_norm=cfg.use_spectral_norm, interp=cfg.interp,
        **({"upsample_v_conv": True} if getattr(cfg, "upsample_v_conv", False) else {}),
        padding_conf={"activate": cfg.pad_activate, "mode": getattr(cfg, "pad_mode", "earth"), "pad_lat": list(cfg.pad_lat),
                      "pad_lon": list(cfg.pad_lon)},
        post_conf=post_conf or {"activate": False})
    sd = synth_state_dict(cfg, family=family)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    return m


def channel_stats(y):
    """per-channel sum, sum of squares, max-abs of [1,C,1,H,W]."""
    a = y[0, :, 0].double()
    return (a.sum(dim=(1, 2)).numpy(), (a * a).sum(dim=(1, 2)).numpy(), a.abs().amax(dim=(1, 2)).numpy())


def model_golden(name, stride, capture_layers, family="base"):
    """family != base: the stress weight families of wxengine.synth -> model_<name>_<family>.npz"""
    cfg = named_config(name)
    torch.manual_seed(0)
    m = reference_model(cfg, family=family)
    x = torch.from_numpy(synth_input(cfg))
    caps = {}
    hooks = []
    if capture_layers:
        want = {f"layers.{s}.0" for s in range(4)} | {f"layers.{s}.1" for s in range(4)}
        want |= {"up_block1", "up_block2", "up_block3", "up_block4"}
        for n, mod in m.named_modules():
            if n in want:
                hooks.append(mod.register_forward_hook(
                    lambda _m, _i, o, n=n: caps.__setitem__(n, o.detach().clone())))
    t = time.time()
    with torch.no_grad():
        y = m(x)
    dt = time.time() - t
    for h in hooks:
        h.remove()
    out = {}
    s1, s2, mx = channel_stats(y)
    out["ch_sum"], out["ch_sumsq"], out["ch_maxabs"] = s1, s2, mx
    out["stride"] = np.int64(stride)
    out["y"] = y[0, :, 0, ::stride, ::stride].numpy().astype(np.float32)
    for n, v in caps.items():
        out["cap/" + n] = v[0].numpy().astype(np.float32)
    if family != "base":
        # what bf16 ARITHMETIC costs on this family, measured on the reference itself: the same module under torch.autocast(bfloat16)
        # against its own fp32 output.  The bf16 engine is gated against this number (it must not be worse than the framework's own bf16).
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            y16 = m(x).float()
        out["bf16_autocast_l2"] = np.float64((y16 - y).norm() / y.norm())
        print(f"[golden] {name} ({family}): reference under torch.autocast(bf16) vs its own fp32: rel-L2 {out['bf16_autocast_l2']:.3e}")
    np.savez_compressed(os.path.join(GOLD, f"model_{name}.npz" if family == "base" else f"model_{name}_{family}.npz"), **out)
    print(f"[golden] {name} ({family}): forward {dt:.2f}s  mean|y|={y.abs().mean():.4f} max|y|={y.abs().max():.4f}  "
          f"y sample {out['y'].shape}")


def layout_golden():
    """build_channel_layout + update_x (credit/datasets/gen_2/channel_utils.py:161-291) on a TWO-source config: the prognostic /
    static / forcing channels of the sources interleave in x, and y carries each source's diagnostics after its prognostics."""
    from credit.datasets.gen_2.channel_utils import build_channel_layout, update_x
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth_batches import two_source_conf
    conf = two_source_conf()
    groups, n_pred = build_channel_layout(conf)
    code = {"prognostic": 0, "dynamic_forcing": 1, "static": 2}
    rows = [(code[g.field_type], g.x_slice.start, -1 if g.src_slice is None else g.src_slice.start, g.x_slice.stop - g.x_slice.start)
            for g in groups.values()]
    c_in = max(g.x_slice.stop for g in groups.values())
    g = np.random.Generator(np.random.Philox(key=[11, 3]))
    x = torch.from_numpy(g.standard_normal((1, c_in, 1, 5, 6), dtype=np.float32))
    frc = torch.from_numpy(g.standard_normal((1, 3, 1, 5, 6), dtype=np.float32))
    y = torch.from_numpy(g.standard_normal((1, 11, 1, 5, 6), dtype=np.float32))
    xn = update_x(x, frc, y, groups)
    np.savez_compressed(os.path.join(GOLD, "channel_layout_two_sources.npz"), groups=np.array(rows, dtype=np.int64), n_pred=np.int64(n_pred),
                        keys=np.array(list(groups.keys())), x=x.numpy(), frc=frc.numpy(), y=y.numpy(), x_new=xn.numpy())
    print(f"[golden] layout: {len(rows)} groups, n_pred {n_pred}, c_in {c_in}: {rows}")


def pad_golden():
    """credit/boundary_padding.py earth mode, incl. asymmetric pads (tests/test_bondary_padding.py:37-44 shapes)."""
    from credit.boundary_padding import TensorPadding
    out = {}
    g = np.random.Generator(np.random.Philox(key=[7, 7]))
    small = torch.from_numpy(g.standard_normal((1, 2, 1, 7, 10), dtype=np.float32))
    tp = TensorPadding(mode="earth", pad_lat=[3, 2], pad_lon=[4, 3])
    out["small_x"] = small.numpy()
    out["small_pad"] = tp.pad(small).numpy()
    big = torch.from_numpy(g.standard_normal((1, 3, 1, 181, 360), dtype=np.float32))
    tp2 = TensorPadding(mode="earth", pad_lat=[12, 34], pad_lon=[56, 78])
    pb = tp2.pad(big)
    assert torch.equal(tp2.unpad(pb), big)
    out["big_seed"] = np.array([7, 7])
    out["big_pad_strided"] = pb[0, :, 0, ::7, ::11].numpy()
    out["big_pad_sum"] = pb.double().sum(dim=(0, 2, 3, 4)).numpy()
    out["big_pad_shape"] = np.array(pb.shape)
    # mode "mirror" (boundary_padding.py:98-134) on the same shapes
    tm = TensorPadding(mode="mirror", pad_lat=[3, 2], pad_lon=[4, 3])
    out["small_mirror"] = tm.pad(small).numpy()
    tm2 = TensorPadding(mode="mirror", pad_lat=[12, 34], pad_lon=[56, 78])
    pm = tm2.pad(big)
    assert torch.equal(tm2.unpad(pm), big)
    out["big_mirror_strided"] = pm[0, :, 0, ::7, ::11].numpy()
    out["big_mirror_sum"] = pm.double().sum(dim=(0, 2, 3, 4)).numpy()
    np.savez_compressed(os.path.join(GOLD, "earth_pad.npz"), **out)
    print("[golden] earth_pad + mirror:", tuple(pb.shape), tuple(pm.shape))


def glue_conf(cfg, n_static=2, n_dyn=2):
    """Minimal data config understood by build_channel_layout (single source)."""
    n_diag = cfg.output_only_channels
    return {
        "data": {"source": {"ERA5": {
            "levels": list(range(cfg.levels)),
            "variables": {
                "prognostic": {"vars_3D": [f"p3_{i}" for i in range(cfg.channels)],
                               "vars_2D": [f"p2_{i}" for i in range(cfg.surface_channels)]},
                "static": {"vars_2D": [f"s_{i}" for i in range(n_static)]},
                "dynamic_forcing": {"vars_2D": [f"f_{i}" for i in range(n_dyn)]},
                "diagnostic": {"vars_2D": [f"d_{i}" for i in range(n_diag)]},
            }}}},
        "model": {"levels": cfg.levels},
    }


def glue_golden():
    """3-step rollout on T0 through the reference's own pieces: model forward, TracerFixer
    (credit/postblock/gen1.py:111, denorm False so no scaler files are needed), y*std+mean
    (rollout_to_netcdf.py:287) and update_x (channel_utils.py:253)."""
    from credit.datasets.gen_2.channel_utils import build_channel_layout, update_x
    from credit.postblock.gen1 import TracerFixer
    cfg = named_config("T0")
    m = reference_model(cfg)
    conf = glue_conf(cfg)
    groups, n_pred = build_channel_layout(conf)
    assert n_pred == cfg.channels * cfg.levels + cfg.surface_channels
    q_inds = list(range(3 * cfg.levels, 4 * cfg.levels))  # 4th 3-D variable = the tracer
    thres = [-0.05] * len(q_inds)  # normalised-space threshold that actually clamps
    fixer = TracerFixer({"tracer_fixer": {"tracer_inds": q_inds, "tracer_thres": thres, "denorm": False}})
    mean, std = synth_denorm(cfg.base_output_channels)
    mean_t = torch.from_numpy(mean).view(1, -1, 1, 1, 1)
    std_t = torch.from_numpy(std).view(1, -1, 1, 1, 1)
    x = torch.from_numpy(synth_input(cfg))
    out = {"tracer_inds": np.array(q_inds), "tracer_thres": np.array(thres, dtype=np.float32),
           "n_static": np.int64(2), "n_dyn": np.int64(2)}
    with torch.no_grad():
        for step in range(1, 4):
            y = m(x)
            n_clamped = int((y[:, q_inds] < thres[0]).sum())
            y = fixer({"y_pred": y, "x": x})["y_pred"]
            y_phys = (y * std_t + mean_t).squeeze(2)
            out[f"y{step}"] = y[0, :, 0].numpy().astype(np.float32)
            out[f"yphys{step}"] = y_phys[0].numpy().astype(np.float32)
            frc = torch.from_numpy(synth_forcing(cfg, 2, step))
            x = update_x(x, frc, y.detach(), groups)
            out[f"x{step}"] = x[0, :, 0].numpy().astype(np.float32)
            print(f"[golden] glue step {step}: clamped {n_clamped} tracer cells, mean|y|={y.abs().mean():.4f}")
    np.savez_compressed(os.path.join(GOLD, "rollout_T0.npz"), **out)


def long_rollout_golden(name, n_steps, stride, with_fp64=True, family="base", dense_steps=(), dense_stride=16):
    """Row R of SURVEY.md 8(a): the predict() loop (rollout_to_netcdf.py:262-316) for n_steps steps on a full-size grid, through the
    reference's own pieces -- CrossFormer forward (fp32, CPU), TracerFixer, y*std+mean, update_x -- and, beside it, the same
    trajectory from the fp64 oracle (oracle/wxformer_oracle.py::rollout).  Stored per step: strided samples of the
    normalised output of both, and per-channel (sum, sum of squares) of the reference output.  The distance between the two
    trajectories is the noise floor of the reference's OWN arithmetic: what an engine can be asked to match at step t."""
    from credit.datasets.gen_2.channel_utils import build_channel_layout, update_x
    from credit.postblock.gen1 import TracerFixer
    from oracle import wxformer_oracle as O
    cfg = named_config(name)
    m = reference_model(cfg, family=family)
    sd = synth_state_dict(cfg, family=family)
    groups, n_pred = build_channel_layout(glue_conf(cfg))
    q_inds = list(range(3 * cfg.levels, 4 * cfg.levels))
    thres = [-0.05] * len(q_inds)
    fixer = TracerFixer({"tracer_fixer": {"tracer_inds": q_inds, "tracer_thres": thres, "denorm": False}})
    x = torch.from_numpy(synth_input(cfg))
    x64 = x.double()
    x16 = x.clone()   # stress families: the reference's own loop under torch.autocast(bfloat16) -- what bf16 arithmetic costs per step
    ac = []
    out = {"tracer_inds": np.array(q_inds), "tracer_thres": np.array(thres, dtype=np.float32), "n_static": np.int64(2),
           "n_dyn": np.int64(2), "stride": np.int64(stride), "n_steps": np.int64(n_steps)}
    ys, y64s, sums, rel, dense = [], [], [], [], []
    t0 = time.time()
    # a 40-step 0.25-degree trajectory is ~35 min of CPU: checkpointed after every step, so an interrupted run resumes where it stopped
    ckpt = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"wx_rollout_golden_{name}_{family}_{n_steps}_{stride}.pt")
    first = 1
    if os.path.isfile(ckpt):
        st = torch.load(ckpt, weights_only=False)
        first, x, x64, x16, ys, y64s, sums, rel, dense, ac = st["step"] + 1, st["x"], st["x64"], st["x16"], st["ys"], st["y64s"], st["sums"], st["rel"], st["dense"], st["ac"]
        print(f"[golden] {name} ({family}) rollout: resuming at step {first} from {ckpt}", flush=True)
    with torch.no_grad():
        for step in range(first, n_steps + 1):
            frc = torch.from_numpy(synth_forcing(cfg, 2, step))
            y = fixer({"y_pred": m(x), "x": x})["y_pred"]
            x = update_x(x, frc, y.detach(), groups)
            if with_fp64:
                y64 = O.tracer_fix(O.forward(cfg, sd, x64, dtype=torch.float64), q_inds, thres)
                x64 = O.update_x(x64, frc.double(), y64, n_pred, 2)
            else:   # 0.25-degree grid: torch's fp64 CPU convolution of the k = 32 CrossEmbed branch wants 157 GB of scratch
                y64 = y.double()
            if family != "base":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    y16 = m(x16).float()
                y16 = fixer({"y_pred": y16, "x": x16})["y_pred"]
                x16 = update_x(x16, frc, y16.detach(), groups)
                ac.append(float((y16 - y).norm() / y.norm()))
            ys.append(y[0, :, 0, ::stride, ::stride].numpy().astype(np.float32))
            y64s.append(y64[0, :, 0, ::stride, ::stride].numpy().astype(np.float32))
            if step in dense_steps:   # a denser sample of a few steps (the strided one is one pixel in 1 600)
                dense.append(y[0, :, 0, ::dense_stride, ::dense_stride].numpy().astype(np.float32))
            s1, s2, _ = channel_stats(y)
            sums.append(np.stack([s1, s2]))
            d = (y.double() - y64)[0, :, 0]
            rel.append(float(d.norm() / y64[0, :, 0].norm()))
            torch.save({"step": step, "x": x, "x64": x64, "x16": x16, "ys": ys, "y64s": y64s, "sums": sums, "rel": rel, "dense": dense, "ac": ac}, ckpt + ".tmp")
            os.replace(ckpt + ".tmp", ckpt)
            print(f"[golden] {name} ({family}) rollout step {step}/{n_steps}: mean|y|={y.abs().mean():.4f}  reference-fp32 vs fp64 oracle "
                  f"rel-L2 {rel[-1]:.3e}" + (f"  reference under bf16 autocast {ac[-1]:.3e}" if ac else "") + f"  ({time.time() - t0:.0f}s)", flush=True)
    out["y"] = np.stack(ys)            # [n_steps, C_out, H/stride, W/stride]  reference, fp32
    # the fp64 oracle's trajectory at the same points (without it: an empty array -- tests then gate against the reference alone)
    out["y64"] = np.stack(y64s) if with_fp64 else np.zeros((0,), np.float32)
    out["ch_sums"] = np.stack(sums)    # [n_steps, 2, C_out] float64
    if dense:
        out["y_dense"], out["dense_steps"], out["dense_stride"] = np.stack(dense), np.array(list(dense_steps)), np.int64(dense_stride)
    out["ref_vs_fp64_rel_l2"] = np.array(rel) if with_fp64 else np.full(n_steps, np.nan)
    if ac:
        out["bf16_autocast_l2"] = np.array(ac)
    np.savez_compressed(os.path.join(GOLD, f"rollout_{name}.npz" if family == "base" else f"rollout_{name}_{family}.npz"), **out)
    if os.path.isfile(ckpt):
        os.remove(ckpt)


def swin_golden():
    """the shifted-window attention of credit/models/swin.py, run through the reference's OWN code where it is
    importable here: `SwinTransformerV2CrBlock._make_attention_mask` (:411-427) and `._shifted_window_attn` (:451-486) called on a
    duck-typed `self`, `window_partition / window_reverse` (:88-118) and `WindowMultiHeadAttention.forward` (:299-330: qkv Linear,
    scaled cosine attention, logit-scale clamp, mask, softmax, P V).  What is NOT reference code: the position-bias table fed into
    that forward -- the class computes it with `timm.layers.Mlp`, and timm is neither vendored nor pinned by the reference
    (SURVEY.md 8(c)), so the table is an input of the fixture (seeded) and the meta-MLP restatement in oracle/swin_oracle.py
    stays unpinned.  Stored: inputs, the seam mask, and the attention core's output (the tensor entering `proj`), image layout."""
    import types
    from credit.models import swin as R
    out = {}
    cases = {"rect_shift": ((12, 16), (4, 8), (2, 4), 2, 32), "fuxi_like": ((14, 21), (7, 7), (3, 3), 1, 128),
             "rect_noshift": ((12, 16), (4, 8), (0, 0), 2, 32), "lat_only_shift": ((16, 12), (8, 4), (5, 0), 3, 32)}
    for name, (feat, ws, shift, heads, hd) in cases.items():
        C = heads * hd
        g = torch.Generator().manual_seed(abs(hash(name)) % 2 ** 31 if False else sum(map(ord, name)))
        attn = R.WindowMultiHeadAttention.__new__(R.WindowMultiHeadAttention)
        torch.nn.Module.__init__(attn)
        attn.in_features, attn.window_size, attn.num_heads, attn.sequential_attn = C, ws, heads, False
        attn.qkv = torch.nn.Linear(C, 3 * C)
        attn.proj = torch.nn.Linear(C, C)
        attn.attn_drop = torch.nn.Identity()
        attn.proj_drop = torch.nn.Identity()
        with torch.no_grad():
            attn.qkv.weight.copy_(torch.randn(3 * C, C, generator=g) / C ** 0.5)
            attn.qkv.bias.copy_(torch.randn(3 * C, generator=g) * 0.1)
        attn.logit_scale = torch.nn.Parameter(torch.log(10 * torch.ones(heads)) + torch.randn(heads, generator=g) * 0.3)
        n = ws[0] * ws[1]
        bias = torch.randn(heads, n, n, generator=g) * 0.5
        attn._relative_positional_encodings = lambda b=bias: b.unsqueeze(0)     # the timm-Mlp part: supplied, see the docstring
        blk = types.SimpleNamespace(feat_size=feat, window_size=ws, shift_size=shift, window_area=n, attn=attn)
        blk.register_buffer = lambda k, v, persistent=False, b=blk: setattr(b, k, v)
        R.SwinTransformerV2CrBlock._make_attention_mask(blk)
        captured = {}
        h = attn.proj.register_forward_pre_hook(lambda _m, inp: captured.__setitem__("core", inp[0].detach().clone()))
        x = torch.randn(1, feat[0], feat[1], C, generator=g)
        with torch.no_grad():
            R.SwinTransformerV2CrBlock._shifted_window_attn(blk, x)
        h.remove()
        core = captured["core"].view(-1, ws[0], ws[1], C)                           # windows of the ROLLED map
        core = R.window_reverse(core, ws, feat)
        if any(shift):
            core = torch.roll(core, shifts=shift, dims=(1, 2))
        out[f"{name}/x"] = x[0].numpy().astype(np.float32)
        out[f"{name}/qkv_w"] = attn.qkv.weight.detach().numpy().astype(np.float32)
        out[f"{name}/qkv_b"] = attn.qkv.bias.detach().numpy().astype(np.float32)
        out[f"{name}/bias"] = bias.numpy().astype(np.float32)
        out[f"{name}/logit_scale_raw"] = attn.logit_scale.detach().numpy().astype(np.float32)
        out[f"{name}/core"] = core[0].numpy().astype(np.float32)
        out[f"{name}/mask"] = (blk.attn_mask.numpy().astype(np.float32) if blk.attn_mask is not None else np.zeros((0,), np.float32))
        out[f"{name}/geom"] = np.array([feat[0], feat[1], ws[0], ws[1], shift[0], shift[1], heads, hd], dtype=np.int64)
        print(f"[golden] swin {name}: feat {feat} ws {ws} shift {shift} heads {heads} hd {hd}  mean|core|={core.abs().mean():.4f}")
    np.savez_compressed(os.path.join(GOLD, "swin_attention.npz"), **out)


def swin_block_golden():
    """SURVEY.md 8(f) row 4, part 2: the whole Swin V2 (Cr) BLOCK and a two-block stage, through the reference's own code:
    `SwinTransformerV2CrBlock.forward` (:484-502), `._shifted_window_attn` (:451-486), `._make_attention_mask` (:411-427) and
    `._calc_window_shift` (:405-409) called on a duck-typed `self` that carries real `nn.Linear` / `nn.LayerNorm` modules, and
    `WindowMultiHeadAttention.forward`, `._relative_positional_encodings`, `._make_pair_wise_relative_positions` (:254-330) on a
    `WindowMultiHeadAttention` built without its constructor.  NOT reference code: `timm.layers.Mlp` (the block's `mlp` and the
    attention's `meta_mlp`) -- timm is neither vendored nor pinned; `_Mlp` below restates it as fc1 -> act -> fc2
    (eval mode: its dropouts are identities), which is the part of this fixture that stays unpinned.
    Blocks alternate shift 0 / window // 2 exactly as SwinTransformerV2CrStage (:616-640) builds them."""
    import types
    from credit.models import swin as R

    class _Mlp(torch.nn.Module):
        def __init__(self, i, h, o, act):
            super().__init__()
            self.fc1, self.act, self.fc2 = torch.nn.Linear(i, h), act(), torch.nn.Linear(h, o)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    out = {}
    cases = {"rect": ((12, 16), (4, 8), 2, 32, 2), "fuxi_like": ((14, 21), (7, 7), 1, 128, 2), "clipped": ((4, 16), (8, 8), 2, 32, 2)}
    for name, (feat, ws_t, heads, hd, depth) in cases.items():
        C = heads * hd
        g = torch.Generator().manual_seed(1000 + sum(map(ord, name)))
        rnd = lambda *shape, s=1.0: torch.randn(*shape, generator=g) * s  # noqa: E731
        x = rnd(1, feat[0], feat[1], C)
        sd = {}
        ys = []
        cur = x
        for i in range(depth):
            blk = types.SimpleNamespace(dim=C, feat_size=feat, target_shift_size=tuple(0 if i % 2 == 0 else w // 2 for w in ws_t))
            blk.window_size, blk.shift_size = R.SwinTransformerV2CrBlock._calc_window_shift(blk, ws_t)
            ws = blk.window_size
            blk.window_area = ws[0] * ws[1]
            attn = R.WindowMultiHeadAttention.__new__(R.WindowMultiHeadAttention)
            torch.nn.Module.__init__(attn)
            attn.in_features, attn.window_size, attn.num_heads, attn.sequential_attn = C, ws, heads, False
            attn.qkv, attn.proj = torch.nn.Linear(C, 3 * C), torch.nn.Linear(C, C)
            attn.attn_drop, attn.proj_drop = torch.nn.Identity(), torch.nn.Identity()
            attn.meta_mlp = _Mlp(2, 48, heads, torch.nn.ReLU)
            attn.logit_scale = torch.nn.Parameter(torch.log(10 * torch.ones(heads)) + rnd(heads, s=0.3))
            R.WindowMultiHeadAttention._make_pair_wise_relative_positions(attn)
            blk.attn = attn
            blk.norm1, blk.norm2, blk.norm3 = torch.nn.LayerNorm(C), torch.nn.LayerNorm(C), torch.nn.Identity()
            blk.drop_path1, blk.drop_path2 = torch.nn.Identity(), torch.nn.Identity()
            blk.mlp = _Mlp(C, 4 * C, C, torch.nn.GELU)
            with torch.no_grad():
                for lin, sc in ((attn.qkv, C ** -0.5), (attn.proj, C ** -0.5), (blk.mlp.fc1, C ** -0.5), (blk.mlp.fc2, (4 * C) ** -0.5),
                                (attn.meta_mlp.fc1, 0.7), (attn.meta_mlp.fc2, 0.15)):
                    lin.weight.copy_(rnd(*lin.weight.shape, s=sc))
                    lin.bias.copy_(rnd(*lin.bias.shape, s=0.1))
                for nm in (blk.norm1, blk.norm2):
                    nm.weight.copy_(1.0 + rnd(C, s=0.2))
                    nm.bias.copy_(rnd(C, s=0.1))
            blk.register_buffer = lambda k, v, persistent=False, b=blk: setattr(b, k, v)
            R.SwinTransformerV2CrBlock._make_attention_mask(blk)
            blk._shifted_window_attn = types.MethodType(R.SwinTransformerV2CrBlock._shifted_window_attn, blk)
            with torch.no_grad():
                cur = R.SwinTransformerV2CrBlock.forward(blk, cur)
            ys.append(cur[0].numpy().astype(np.float32))
            for mod, pre in ((attn.qkv, "attn.qkv."), (attn.proj, "attn.proj."), (attn.meta_mlp.fc1, "attn.meta_mlp.fc1."),
                             (attn.meta_mlp.fc2, "attn.meta_mlp.fc2."), (blk.norm1, "norm1."), (blk.norm2, "norm2."),
                             (blk.mlp.fc1, "mlp.fc1."), (blk.mlp.fc2, "mlp.fc2.")):
                sd[f"blocks.{i}.{pre}weight"] = mod.weight.detach().numpy().astype(np.float32)
                sd[f"blocks.{i}.{pre}bias"] = mod.bias.detach().numpy().astype(np.float32)
            sd[f"blocks.{i}.attn.logit_scale"] = attn.logit_scale.detach().numpy().astype(np.float32)
        out[f"{name}/x"] = x[0].numpy().astype(np.float32)
        out[f"{name}/geom"] = np.array([feat[0], feat[1], ws_t[0], ws_t[1], heads, hd, depth], dtype=np.int64)
        for k, v in sd.items():
            out[f"{name}/sd/{k}"] = v
        for i, y in enumerate(ys):
            out[f"{name}/y{i}"] = y
        print(f"[golden] swin block {name}: feat {feat} window {ws_t} heads {heads} hd {hd} depth {depth}  mean|y|={np.abs(ys[-1]).mean():.4f}")
    np.savez_compressed(os.path.join(GOLD, "swin_block.npz"), **out)


def attend_golden():
    """SURVEY.md 8(f) row 4, `Attend` (credit/attend.py:94-120, the non-flash branch: softmax(q k^T * scale) v per batch item and
    head) run through the reference class itself.  Stored: q, k, v [b, h, n, d] and the output, two shapes / both scale conventions."""
    from credit.attend import Attend
    out = {}
    for name, (b, h, n, d, scale) in {"n64_d32": (3, 4, 64, 32, None), "n128_d64_scaled": (2, 2, 128, 64, 0.2), "n100_d32": (2, 4, 100, 32, None)}.items():
        g = torch.Generator().manual_seed(sum(map(ord, name)))
        q, k, v = (torch.randn(b, h, n, d, generator=g) for _ in range(3))
        att = Attend(dropout=0.0, flash=False, scale=scale).eval()
        with torch.no_grad():
            o = att(q, k, v)
        for nm, t in (("q", q), ("k", k), ("v", v), ("out", o)):
            out[f"{name}/{nm}"] = t.numpy().astype(np.float32)
        out[f"{name}/scale"] = np.array([np.nan if scale is None else scale], dtype=np.float32)
        print(f"[golden] attend {name}: {tuple(q.shape)} scale {scale} mean|out|={o.abs().mean():.4f}")
    np.savez_compressed(os.path.join(GOLD, "attend.npz"), **out)


def fuxi_golden():
    """BASELINE config 5: the FuXi forward through the reference's own modules (credit/models/fuxi.py): `CubeEmbedding`, `DownBlock`,
    `UpBlock`, `get_pad2d`, `UTransformer.forward`, `Fuxi.forward`, `apply_spectral_norm`, the two classes built without their
    constructors (which need timm) and populated with exactly the attributes those constructors set.  In the place of timm's
    `SwinTransformerV2Stage` (not vendored, not pinned: SURVEY.md 8(c)) sits a stage of the reference's OWN V2-Cr blocks
    (credit/models/swin.py: `SwinTransformerV2CrBlock.forward / _shifted_window_attn / _make_attention_mask / _calc_window_shift`,
    `WindowMultiHeadAttention.forward`), shift alternating 0 / window // 2 as SwinTransformerV2CrStage (:616-640) builds them; `_Mlp`
    restates timm.layers.Mlp (fc1 -> act -> fc2).  Weights: wxengine.fuxi.synth_fuxi_state_dict, loaded strict=True -- which also
    pins the key names and shapes of FuxiConfig.state_spec() outside the stage.  Stored: y and four intermediate maps."""
    from credit.models import fuxi as RF
    from credit.models import swin as RS
    from wxengine.fuxi import named_fuxi_config, synth_fuxi_state_dict
    from wxengine.synth import keyed_normal
    nn = torch.nn

    class _Mlp(nn.Module):
        def __init__(self, i, h, o, act):
            super().__init__()
            self.fc1, self.act, self.fc2 = nn.Linear(i, h), act(), nn.Linear(h, o)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class _CrBlock(nn.Module):
        forward = RS.SwinTransformerV2CrBlock.forward
        _shifted_window_attn = RS.SwinTransformerV2CrBlock._shifted_window_attn
        _make_attention_mask = RS.SwinTransformerV2CrBlock._make_attention_mask
        _calc_window_shift = RS.SwinTransformerV2CrBlock._calc_window_shift

        def __init__(self, C, feat, ws_t, heads, shifted, meta_hidden):
            super().__init__()
            self.dim, self.feat_size = C, feat
            self.target_shift_size = tuple(w // 2 if shifted else 0 for w in ws_t)
            self.window_size, self.shift_size = self._calc_window_shift(ws_t)
            self.window_area = self.window_size[0] * self.window_size[1]
            attn = RS.WindowMultiHeadAttention.__new__(RS.WindowMultiHeadAttention)
            nn.Module.__init__(attn)
            attn.in_features, attn.window_size, attn.num_heads, attn.sequential_attn = C, self.window_size, heads, False
            attn.qkv, attn.proj = nn.Linear(C, 3 * C), nn.Linear(C, C)
            attn.attn_drop, attn.proj_drop = nn.Identity(), nn.Identity()
            attn.meta_mlp = _Mlp(2, meta_hidden, heads, nn.ReLU)
            attn.logit_scale = nn.Parameter(torch.zeros(heads))
            attn._make_pair_wise_relative_positions()
            self.attn = attn
            self.norm1, self.norm2, self.norm3 = nn.LayerNorm(C), nn.LayerNorm(C), nn.Identity()
            self.drop_path1, self.drop_path2 = nn.Identity(), nn.Identity()
            self.mlp = _Mlp(C, 4 * C, C, nn.GELU)
            self._make_attention_mask()

    class _CrStage(nn.Module):
        def __init__(self, C, feat, window, heads, depth, meta_hidden):
            super().__init__()
            self.blocks = nn.ModuleList([_CrBlock(C, feat, (window, window), heads, i % 2 == 1, meta_hidden) for i in range(depth)])

        def forward(self, x):
            for b in self.blocks:
                x = b(x)
            return x

    for name in ("FT0", "FT1", "FT2"):
        cfg = named_fuxi_config(name)
        img = (cfg.frames, cfg.image_height, cfg.image_width)
        patch = (cfg.frame_patch_size, cfg.patch_height, cfg.patch_width)
        m = RF.Fuxi.__new__(RF.Fuxi)
        nn.Module.__init__(m)
        # the attributes Fuxi.__init__ (:358-452) sets, padding_conf / post_conf / noise off
        m.use_interp, m.use_spectral_norm, m.use_padding, m.use_post_block = cfg.interp, cfg.use_spectral_norm, False, False
        m.img_size_original = m.img_size = img
        m.patch_size = patch
        m.input_resolution = (round(img[1] / patch[1] / 2), round(img[2] / patch[2] / 2))
        m.out_chans = cfg.out_chans
        m.cube_embedding = RF.CubeEmbedding(img, patch, cfg.in_chans, cfg.dim)
        ut = RF.UTransformer.__new__(RF.UTransformer)
        nn.Module.__init__(ut)
        # UTransformer.__init__ (:216-276)
        ut.padding = RF.get_pad2d(m.input_resolution, (cfg.window_size, cfg.window_size))
        ut.pad = nn.ZeroPad2d(ut.padding)
        res = (m.input_resolution[0] + ut.padding[2] + ut.padding[3], m.input_resolution[1] + ut.padding[0] + ut.padding[1])
        assert res == cfg.stage_feat, (res, cfg.stage_feat)
        ut.down = RF.DownBlock(cfg.dim, cfg.dim, cfg.groups[0])
        ut.layer = _CrStage(cfg.dim, res, cfg.window_size, cfg.num_heads, cfg.depth, cfg.meta_hidden)
        ut.up = RF.UpBlock(cfg.dim * 2, cfg.dim, cfg.groups[1])
        ut.use_noise = False
        m.u_transformer = ut
        m.fc = nn.Linear(cfg.dim, cfg.out_chans * patch[1] * patch[2])
        if cfg.use_spectral_norm:
            RF.apply_spectral_norm(m)
        sd = synth_fuxi_state_dict(cfg)
        res_ld = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        assert not res_ld.missing_keys and not res_ld.unexpected_keys
        m.eval()
        taps = {}
        hooks = [m.cube_embedding.register_forward_hook(lambda _m, _i, o: taps.__setitem__("embed", o[0, :, 0].permute(1, 2, 0))),
                 ut.down.register_forward_hook(lambda _m, _i, o: taps.__setitem__("down", o[0].permute(1, 2, 0))),
                 ut.layer.register_forward_hook(lambda _m, _i, o: taps.__setitem__("stage_padded", o[0])),
                 ut.register_forward_hook(lambda _m, _i, o: taps.__setitem__("up", o[0].permute(1, 2, 0)))]
        x = torch.from_numpy(keyed_normal("fuxi/x0", (1, cfg.in_chans, cfg.frames, cfg.image_height, cfg.image_width), 1000))
        with torch.no_grad():
            y = RF.Fuxi.forward(m, x)
        for h in hooks:
            h.remove()
        pl, _, pt, _ = ut.padding
        hd, wd = m.input_resolution
        taps["stage"] = taps.pop("stage_padded")[pt: pt + hd, pl: pl + wd]
        assert y.shape == (1, cfg.out_chans, 1, cfg.image_height, cfg.image_width)
        out = {"y": y[0, :, 0].numpy().astype(np.float32), "padding": np.array(ut.padding, dtype=np.int64)}
        for k, v in taps.items():
            out[k] = v.contiguous().numpy().astype(np.float32)
        np.savez_compressed(os.path.join(GOLD, f"fuxi_{name}.npz"), **out)
        print(f"[golden] fuxi {name}: in {tuple(x.shape)} stage map {res} padding {ut.padding}  mean|y|={y.abs().mean():.4f} "
              + " ".join(f"{k}:{tuple(v.shape)}" for k, v in taps.items()))


def fixer_inputs(seed=11):
    """Physically plausible random fields on the reference's 10x18 / 7-level demo grid.
    x: [T(7) | q(7) | U(7) | V(7)] x 2 frames; y: the same 28 + [TOA solar, TOA OLR, surf solar, surf LR, SH, LH, precip, evapor]."""
    g = np.random.Generator(np.random.Philox(key=[seed, 3]))
    H, W, L = 10, 18, 7

    def state():
        T = 250.0 + 30.0 * g.standard_normal((L, H, W))
        q = np.abs(0.004 + 0.004 * g.standard_normal((L, H, W)))
        U = 12.0 * g.standard_normal((L, H, W))
        V = 8.0 * g.standard_normal((L, H, W))
        return np.concatenate([T, q, U, V], 0)
    x = np.stack([state(), state()], 1).astype(np.float32)            # [28, 2, H, W]
    flux = 2.0e6 * g.standard_normal((6, H, W))
    precip = np.abs(2e-3 * g.standard_normal((1, H, W)))
    evapor = -np.abs(1e-3 * g.standard_normal((1, H, W)))
    y = np.concatenate([state() * 1.0, flux, precip, evapor], 0).astype(np.float32)  # [36, H, W]
    return x, y


def fixers_golden():
    """GlobalMassFixer / GlobalWaterFixer / GlobalEnergyFixer of the reference on its own simple_demo grid
    (credit/postblock/gen1.py:188-223), trapz and midpoint, denorm False."""
    from credit.postblock.gen1 import GlobalEnergyFixer, GlobalMassFixer, GlobalWaterFixer
    x_np, y_np = fixer_inputs()
    out = {"x": x_np, "y": y_np}
    L = 7
    for midpoint in (False, True):
        nl = L - 1 if midpoint else L      # midpoint variants take one level fewer (mid-level values)
        tag = "mid" if midpoint else "trapz"
        # channel layout for this variant: first nl levels of each 7-level block
        xs = np.concatenate([x_np[b * L:b * L + nl] for b in range(4)], 0)
        ys = np.concatenate([y_np[b * L:b * L + nl] for b in range(4)] + [y_np[28:]], 0)
        x = torch.from_numpy(xs)[None]
        y = torch.from_numpy(ys)[None, :, None]
        q0 = nl
        base = {"simple_demo": True, "denorm": False, "grid_type": "pressure", "midpoint": midpoint,
                "activate": True, "activate_outside_model": False}
        conf_m = {"global_mass_fixer": dict(base, fix_level_num=3, q_inds=list(range(q0, q0 + nl))),
                  "data": {"lead_time_periods": 6}}
        conf_w = {"global_water_fixer": dict(base, q_inds=list(range(q0, q0 + nl)), precip_ind=4 * nl + 6,
                                             evapor_ind=4 * nl + 7), "data": {"lead_time_periods": 6}}
        conf_e = {"global_energy_fixer": dict(base, T_inds=list(range(0, nl)), q_inds=list(range(q0, q0 + nl)),
                                              U_inds=list(range(2 * nl, 3 * nl)), V_inds=list(range(3 * nl, 4 * nl)),
                                              TOA_rad_inds=[4 * nl, 4 * nl + 1], surf_rad_inds=[4 * nl + 2, 4 * nl + 3],
                                              surf_flux_inds=[4 * nl + 4, 4 * nl + 5]), "data": {"lead_time_periods": 6}}
        with torch.no_grad():
            ym = GlobalMassFixer(conf_m)({"y_pred": y.clone(), "x": x.clone()})["y_pred"]
            yw = GlobalWaterFixer(conf_w)({"y_pred": y.clone(), "x": x.clone()})["y_pred"]
            ye = GlobalEnergyFixer(conf_e)({"y_pred": y.clone(), "x": x.clone()})["y_pred"]
            # chained, in PostBlock order (mass -> water -> energy)
            yc = GlobalEnergyFixer(conf_e)(GlobalWaterFixer(conf_w)(GlobalMassFixer(conf_m)(
                {"y_pred": y.clone(), "x": x.clone()})))["y_pred"]
        for name, t in (("mass", ym), ("water", yw), ("energy", ye), ("chain", yc)):
            assert t.shape == y.shape, (name, t.shape)
            out[f"{tag}_{name}"] = t[0, :, 0].double().numpy()
        print(f"[golden] fixers {tag}: mass dq max {float((ym - y).abs().max()):.3e}  water dP max "
              f"{float((yw - y).abs().max()):.3e}  energy dT max {float((ye - y).abs().max()):.3e}")
    np.savez_compressed(os.path.join(GOLD, "fixers_demo.npz"), **out)


def fixers_updown_golden():
    """GlobalEnergyFixerUpDown of the reference on its simple_demo grid (gen1.py:866-879), trapz and midpoint.
    Channel layout of y: [T|q|U|V] + 9 flux channels [TOA dn solar, TOA up solar, OLR, surf dn solar, surf up solar, surf dn LW,
    surf up LW, SH, LH]."""
    from credit.postblock.gen1 import GlobalEnergyFixerUpDown
    x_np, y_np = fixer_inputs(seed=13)
    g = np.random.Generator(np.random.Philox(key=[13, 9]))
    L, H, W = 7, 10, 18
    flux = np.abs(3.0e6 * g.standard_normal((9, H, W))).astype(np.float32)
    out = {"x": x_np, "y": y_np[:28], "flux": flux}
    for midpoint in (False, True):
        nl = L - 1 if midpoint else L
        tag = "mid" if midpoint else "trapz"
        xs = np.concatenate([x_np[b * L:b * L + nl] for b in range(4)], 0)
        ys = np.concatenate([y_np[b * L:b * L + nl] for b in range(4)] + [flux], 0)
        x = torch.from_numpy(xs)[None]
        y = torch.from_numpy(ys)[None, :, None]
        f0 = 4 * nl
        cfg = {"simple_demo": True, "denorm": False, "grid_type": "pressure", "midpoint": midpoint, "activate": True,
               "activate_outside_model": False, "T_inds": list(range(0, nl)), "q_inds": list(range(nl, 2 * nl)),
               "U_inds": list(range(2 * nl, 3 * nl)), "V_inds": list(range(3 * nl, 4 * nl)),
               "TOA_down_solar_ind": f0, "TOA_up_solar_ind": f0 + 1, "TOA_up_OLR_ind": f0 + 2, "surf_down_solar_ind": f0 + 3,
               "surf_up_solar_ind": f0 + 4, "surf_down_LW_ind": f0 + 5, "surf_up_LW_ind": f0 + 6, "surf_SH_ind": f0 + 7,
               "surf_LH_ind": f0 + 8}
        with torch.no_grad():
            ye = GlobalEnergyFixerUpDown({"global_energy_fixer_updown": cfg, "data": {"lead_time_periods": 6}})(
                {"y_pred": y.clone(), "x": x.clone()})["y_pred"]
        out[f"{tag}_updown"] = ye[0, :, 0].double().numpy()
        print(f"[golden] updown energy fixer {tag}: dT max {float((ye - y).abs().max()):.3e}")
    np.savez_compressed(os.path.join(GOLD, "fixers_updown.npz"), **out)


def preblock_golden():
    """ERA5Normalizer (stats injected, the NetCDF reading bypassed: xarray is absent) -> ConcatToTensor of the reference."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth_batches import preblock_batch
    from credit.preblock.concat import ConcatToTensor
    from credit.preblock.norm import ERA5Normalizer
    batch, mean, std = preblock_batch()
    n = ERA5Normalizer.__new__(ERA5Normalizer)
    torch.nn.Module.__init__(n)
    n._mean = {k: torch.tensor(np.array(v), dtype=torch.float32) for k, v in mean.items()}
    n._std = {k: torch.tensor(np.array(v), dtype=torch.float32) for k, v in std.items()}
    x, meta = ConcatToTensor()(n(batch))
    cmap = meta["input"]["_channel_map"]
    out = {"x": x.numpy(), "keys": np.array(list(cmap.keys())), "starts": np.array([v["slice"].start for v in cmap.values()]),
           "stops": np.array([v["slice"].stop for v in cmap.values()])}
    x2, _ = ConcatToTensor()(batch)   # concatenation only
    out["x_raw"] = x2.numpy()
    np.savez_compressed(os.path.join(GOLD, "preblock.npz"), **out)
    print(f"[golden] preblock: x {tuple(x.shape)} order {list(cmap.keys())}")


def conservation_golden():
    """Gen-2 name-keyed fixers (credit/postblock/conservation.py) on a hybrid sigma grid; `get_forward_data` (xarray) is
    replaced by an in-memory provider of the arrays defined here, the classes and physics core run unmodified."""
    import credit.postblock.conservation as G2
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth_batches import conservation_batch
    lat = np.array([90, 70, 50, 30, 10, -10, -30, -50, -70, -90], dtype=np.float64)
    lon2d, lat2d = np.meshgrid(np.arange(0, 360, 20, dtype=np.float64), lat)

    class _V:
        def __init__(self, a):
            self.values = a
    out = {}
    for midpoint in (True, False):
        tag = "mid" if midpoint else "trapz"
        batch, gph = conservation_batch(midpoint=midpoint)
        fake = {"lon2d": _V(lon2d), "lat2d": _V(lat2d), "coef_a": _V(SIGMA_A), "coef_b": _V(SIGMA_B), "PHIS": _V(gph)}
        G2.get_forward_data = lambda _f: fake
        phys = dict(save_loc_physics="in-memory.nc", lon_lat_level_name=["lon2d", "lat2d", "coef_a", "coef_b"], grid_type="sigma",
                    midpoint=midpoint)
        P = "cam/prognostic/"
        D = "cam/diagnostic/2d/"
        fixers = [
            G2.TracerFixer([P + "3d/Qtot", D + "PRECT"], [1e-9, 0.0], [None, 1.0]),
            G2.GlobalMassFixer(P + "3d/Qtot", P + "2d/PS", **phys),
            G2.GlobalWaterFixer(P + "3d/Qtot", P + "2d/PS", D + "PRECT", D + "QFLX", 6, **phys),
            G2.GlobalEnergyFixerUpDown(P + "3d/T", P + "3d/Qtot", P + "3d/U", P + "3d/V", P + "2d/PS", ["PHIS"],
                                       "cam/dynamic_forcing/2d/SOLIN", D + "FSUTOA", D + "FLUT", D + "FSDS", D + "FSUS", D + "FLDS",
                                       D + "FLUS", D + "SHFLX", D + "LHFLX", 6, **phys)]
        with torch.no_grad():
            for f in fixers:
                batch = f(batch)
        for k in (P + "2d/PS", D + "PRECT", P + "3d/T", P + "3d/Qtot"):
            out[f"{tag}:{k}"] = batch["y_processed"]["cam"][k].numpy()
        print(f"[golden] gen-2 conservation chain {tag}: done")
    np.savez_compressed(os.path.join(GOLD, "conservation_gen2.npz"), **out)


def reconstruct_golden():
    """Reconstruct -> FlattenToTensor of the reference (no scaler) on a synthetic y_pred + channel map."""
    from credit.postblock.reconstruct import FlattenToTensor, Reconstruct
    g = np.random.Generator(np.random.Philox(key=[41, 1]))
    B, H, W = 2, 9, 14
    y = torch.from_numpy(g.standard_normal((B, 11, 1, H, W)).astype(np.float32))
    cmap = {"era5/prognostic/3d/T": {"slice": slice(0, 4), "orig_shape": (4, 1)},
            "era5/prognostic/3d/Q": {"slice": slice(4, 8), "orig_shape": (4, 1)},
            "era5/prognostic/2d/SP": {"slice": slice(8, 9), "orig_shape": (1, 1)},
            "era5/diagnostic/2d/tp": {"slice": slice(9, 10), "orig_shape": (1, 1)},
            "era5/diagnostic/2d/evap": {"slice": slice(10, 11), "orig_shape": (1, 1)}}
    bd = {"y_pred": y.clone(), "metadata": {"target": {"_channel_map": cmap}}}
    bd = Reconstruct()(bd)
    out = {"y": y.numpy()}
    for k, v in bd["y_processed"]["era5"].items():
        out["rec:" + k] = v.numpy()
    bd["y_processed"]["era5"]["era5/prognostic/2d/SP"] = bd["y_processed"]["era5"]["era5/prognostic/2d/SP"] * 2.0
    bd = FlattenToTensor()(bd)
    out["flat"] = bd["y_pred"].numpy()
    np.savez_compressed(os.path.join(GOLD, "reconstruct.npz"), **out)
    print("[golden] reconstruct / flatten:", tuple(bd["y_pred"].shape))


def assemble_golden():
    """assemble_rollout_batch of the reference (rollout_utils.py:322-430): which tensor object each key is routed to."""
    from credit.trainers.rollout_utils import assemble_rollout_batch
    def t(v):
        return torch.full((1, 1, 1, 2, 2), float(v))
    keys_ic = ["era5/prognostic/3d/T", "era5/prognostic/2d/SP", "era5/static/2d/LSM", "era5/dynamic_forcing/2d/tsi",
               "era5/dynamic_forcing/2d/sza", "era5/diagnostic/2d/tp"]
    ic = {"input": {"era5": {k: t(100 + i) for i, k in enumerate(keys_ic)}, "empty": {}}}
    pred = {"era5": {"era5/prognostic/3d/T": t(200), "era5/prognostic/2d/SP": t(201), "era5/diagnostic/2d/tp": t(202)}}
    cur = {"input": {"era5": {"era5/dynamic_forcing/2d/tsi": t(300)}}, "target": None}   # sza missing -> carried forward
    out = assemble_rollout_batch({"y_processed": pred, "ic_preprocessed": ic}, cur, 1)
    res = {"keys": np.array(list(out["input"]["era5"].keys())), "vals": np.array([float(v.flatten()[0]) for v in out["input"]["era5"].values()]),
           "sources": np.array(list(out["input"].keys()))}
    np.savez_compressed(os.path.join(GOLD, "assemble_rollout.npz"), **res)
    print("[golden] assemble_rollout_batch:", dict(zip(res["keys"], res["vals"])))



def gen2loop_golden():
    """The COMPOSED gen-2 loop: the reference's own `run_forecast` (credit/trainers/rollout_utils.py:204-319) driven for three steps on T0
    -- its `apply_preblocks` over the reference's ERA5Normalizer (statistics injected: xarray is absent) + ConcatToTensor (with a
    ChannelSchema, so the diagnostics reach the target channel map as at inference), the reference CrossFormer, its `apply_postblocks`
    over the reference's Reconstruct and an inverse scaler, its `assemble_rollout_batch` between steps.  The inverse scaler is the one
    block that is not reference code (the reference's is a bridgescaler wrapper; bridgescaler is not installable here): y * std + mean
    per variable, written here as a BasePostblock.  Stored: every y_processed variable of every step (stride 2 in both map axes)."""
    import credit.trainers.rollout_utils as RU
    from credit.datasets.gen_2.channel_utils import ChannelSchema
    from credit.postblock.base import BasePostblock
    from credit.postblock.reconstruct import Reconstruct
    from credit.preblock.concat import ConcatToTensor
    from credit.preblock.norm import ERA5Normalizer
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth_batches import gen2loop_batches, gen2loop_schema
    cfg = named_config("T0")
    n_steps = 3
    ic, frcs, mean, std = gen2loop_batches(cfg, n_steps)
    inp, out = gen2loop_schema(cfg)

    norm = ERA5Normalizer.__new__(ERA5Normalizer)
    torch.nn.Module.__init__(norm)
    norm._mean = {k: torch.tensor(np.array(v), dtype=torch.float32) for k, v in mean.items()}
    norm._std = {k: torch.tensor(np.array(v), dtype=torch.float32) for k, v in std.items()}
    concat = ConcatToTensor()
    concat.set_schema(ChannelSchema([{"var_key": k, "n_levels": nl, "n_time": 1} for k, nl in inp],
                                    [{"var_key": k, "n_levels": nl, "n_time": 1} for k, nl in out]))

    class InverseScale(BasePostblock):
        def forward(self, batch_dict):
            for src, variables in batch_dict["y_processed"].items():
                for key in list(variables):
                    name = key.split("/")[-1]
                    if name in mean:
                        variables[key] = variables[key] * torch.as_tensor(std[name]).reshape(1, -1, 1, 1, 1) + torch.as_tensor(mean[name]).reshape(1, -1, 1, 1, 1)
            return batch_dict
    step_pre = torch.nn.ModuleDict({"norm": norm, "concat": concat})
    step_post = torch.nn.ModuleDict({"reconstruct": Reconstruct(), "inverse_scaler": InverseScale()})
    model = reference_model(cfg)
    t0 = np.datetime64("2020-01-01T00").astype("datetime64[ns]").astype(np.int64)
    ic_batch = {"input": ic["input"], "metadata": {"era5": {"input_datetime": torch.tensor([int(t0)])}}}
    RU.decode_time = lambda ns, calendar="standard": __import__("pandas").Timestamp(int(ns))   # cftime is absent; the standard calendar is plain pandas
    got = []

    def save(y_processed, init_time, step, fhr_per_step, save_dir, pool):
        got.append({k: v.clone() for k, v in y_processed["era5"].items()})
    conf = {"data": {"timestep": "6h", "history_len": 1}}
    RU.run_forecast(conf, n_steps, "unused", torch.nn.ModuleDict(), step_pre, step_post, torch.nn.ModuleDict(), model,
                    iter([ic_batch] + frcs), torch.device("cpu"), None, save, verbose=False)
    assert len(got) == n_steps
    res = {"keys": np.array(list(got[0].keys()))}
    for s, d in enumerate(got):
        for k, v in d.items():
            res[f"step{s}:{k}"] = v.numpy()[..., ::2, ::2]
    np.savez_compressed(os.path.join(GOLD, "gen2_loop_T0.npz"), **res)
    print(f"[golden] gen-2 composed loop (reference run_forecast, T0, {n_steps} steps): {len(got[0])} variables per step, "
          f"max |y| {max(float(v.abs().max()) for v in got[-1].values()):.3f}")


def fuxi_timm_golden():
    """BASELINE config 5's default stage variant is timm's SwinTransformerV2Stage, and timm is not installable in the build container:
    the engine's timm block follows timm's published source and is unpinned.  This entry pins it wherever `import timm` works: it builds
    the reference Fuxi (credit/models/fuxi.py) at the FT0 geometry with the real timm stage, loads the build's synthetic weights and
    writes tests/golden/fuxi_timm_FT{0,1,2}T.npz, which tests/test_fuxi.py picks up (and skips, loudly, while the files are absent)."""
    try:
        import timm  # noqa: F401
        if "MagicMock" in type(timm).__name__ or not hasattr(timm, "__version__") or not isinstance(timm.__version__, str):
            raise ImportError("timm is the oracle stub")
    except Exception as e:   # noqa: BLE001
        print(f"[golden] fuxi_timm: SKIPPED -- `import timm` does not give the real package here ({e}); run "
              "`python tools/make_goldens.py --only fuxi_timm` in an environment that has timm to pin BASELINE config 5's default stage")
        return
    from credit.models.fuxi import Fuxi
    from wxengine.fuxi import named_fuxi_config, synth_fuxi_state_dict
    from wxengine.synth import keyed_normal
    for name in ("FT0T", "FT1T", "FT2T"):   # tests/test_fuxi.py's three geometries with the reference's own stage
        cfg = named_fuxi_config(name)
        m = Fuxi(image_height=cfg.image_height, patch_height=cfg.patch_height, image_width=cfg.image_width, patch_width=cfg.patch_width,
                 levels=cfg.levels, frames=cfg.frames, frame_patch_size=cfg.frame_patch_size, dim=cfg.dim, num_groups=cfg.num_groups,
                 channels=cfg.channels, surface_channels=cfg.surface_channels, input_only_channels=cfg.input_only_channels,
                 output_only_channels=cfg.output_only_channels, num_heads=cfg.num_heads, depth=cfg.depth, window_size=cfg.window_size,
                 use_spectral_norm=cfg.use_spectral_norm, interp=cfg.interp)
        sd = synth_fuxi_state_dict(cfg)
        res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        bad = [k for k in list(res.missing_keys) + list(res.unexpected_keys) if not k.endswith(("relative_position_index", "relative_coords_table", "attn_mask"))]
        assert not bad, f"{name}: the synthetic state dict and the reference module disagree on keys: {bad[:6]}"
        m.eval()   # built and loaded on the CPU, never moved or cast: the case FuxiConfig.timm_qkv_unnormalised = True describes
        x = torch.from_numpy(keyed_normal("fuxi/x0", (1, cfg.in_chans, cfg.frames, cfg.image_height, cfg.image_width), 1000))
        with torch.no_grad():
            y = m(x)
        np.savez_compressed(os.path.join(GOLD, f"fuxi_timm_{name}.npz"), y=y[0, :, 0].numpy().astype(np.float32), timm_version=np.array(timm.__version__))
        print(f"[golden] fuxi_timm {name}: y {tuple(y.shape)} mean|y| {float(y.abs().mean()):.4f} (timm {timm.__version__})")


SIGMA_A = np.array([200.0, 5000.0, 12000.0, 14000.0, 9000.0, 3000.0, 0.0], dtype=np.float32)       # Pa
SIGMA_B = np.array([0.0, 0.0, 0.08, 0.3, 0.6, 0.88, 1.0], dtype=np.float32)


def fixers_sigma_golden():
    """The same three fixers on a hybrid sigma-pressure grid (gen1.py sigma branches).  The reference reads its grid
    from `save_loc_physics` through xarray (absent here): `get_forward_data` is replaced by an in-memory stand-in that
    serves the arrays this script defines -- the fixer classes and physics_hybrid_sigma_level run unmodified."""
    import credit.postblock.gen1 as G
    x_np, y_np = fixer_inputs(seed=12)
    g = np.random.Generator(np.random.Philox(key=[12, 5]))
    H, W, L = 10, 18, 7
    sp_x = (1.0e5 + 2.0e3 * g.standard_normal((1, 2, H, W))).astype(np.float32)
    sp_y = (1.0e5 + 2.0e3 * g.standard_normal((1, H, W))).astype(np.float32)
    gph = (50.0 + 20.0 * g.standard_normal((H, W))).astype(np.float32)
    lat = np.array([90, 70, 50, 30, 10, -10, -30, -50, -70, -90], dtype=np.float64)
    lon = np.arange(0, 360, 20, dtype=np.float64)
    lon2d, lat2d = np.meshgrid(lon, lat)

    class _V:
        def __init__(self, a):
            self.values = a
    fake = {"lon2d": _V(lon2d), "lat2d": _V(lat2d), "coef_a": _V(SIGMA_A), "coef_b": _V(SIGMA_B), "gph": _V(gph)}
    G.get_forward_data = lambda _f: fake
    out = {"x": x_np, "y": y_np, "sp_x": sp_x, "sp_y": sp_y, "gph": gph, "coef_a": SIGMA_A, "coef_b": SIGMA_B}
    for midpoint in (False, True):
        nl = L - 1 if midpoint else L
        tag = "mid" if midpoint else "trapz"
        xs = np.concatenate([x_np[b * L:b * L + nl] for b in range(4)] + [sp_x], 0)            # [4 nl + 1, 2, H, W]
        ys = np.concatenate([y_np[b * L:b * L + nl] for b in range(4)] + [y_np[28:], sp_y], 0)  # [4 nl + 8 + 1, H, W]
        # SP sits at the same index in x and y (gen1.py:306-308): pad x with zero channels up to y's SP index
        sp_ind = 4 * nl + 8
        xs = np.concatenate([xs[:4 * nl], np.zeros((8, 2, H, W), np.float32), xs[4 * nl:]], 0)
        x = torch.from_numpy(xs)[None]
        y = torch.from_numpy(ys)[None, :, None]
        base = {"simple_demo": False, "denorm": False, "grid_type": "sigma", "midpoint": midpoint, "activate": True,
                "activate_outside_model": False, "lon_lat_level_name": ["lon2d", "lat2d", "coef_a", "coef_b"],
                "sp_inds": sp_ind}
        q0 = nl
        mass = dict(base, fix_level_num=3, q_inds=list(range(q0, q0 + nl)))
        data = {"lead_time_periods": 6, "save_loc_physics": "in-memory.nc"}
        conf_m = {"global_mass_fixer": mass, "data": data}
        conf_w = {"global_mass_fixer": mass, "data": data,
                  "global_water_fixer": dict(base, q_inds=list(range(q0, q0 + nl)), precip_ind=4 * nl + 6, evapor_ind=4 * nl + 7)}
        conf_e = {"global_mass_fixer": mass, "data": data,
                  "global_energy_fixer": dict(base, T_inds=list(range(0, nl)), q_inds=list(range(q0, q0 + nl)),
                                              U_inds=list(range(2 * nl, 3 * nl)), V_inds=list(range(3 * nl, 4 * nl)),
                                              TOA_rad_inds=[4 * nl, 4 * nl + 1], surf_rad_inds=[4 * nl + 2, 4 * nl + 3],
                                              surf_flux_inds=[4 * nl + 4, 4 * nl + 5], surface_geopotential_name=["gph"])}
        with torch.no_grad():
            ym = G.GlobalMassFixer(conf_m)({"y_pred": y.clone(), "x": x.clone()})["y_pred"]
            yw = G.GlobalWaterFixer(conf_w)({"y_pred": y.clone(), "x": x.clone()})["y_pred"]
            ye = G.GlobalEnergyFixer(conf_e)({"y_pred": y.clone(), "x": x.clone()})["y_pred"]
            yc = G.GlobalEnergyFixer(conf_e)(G.GlobalWaterFixer(conf_w)(G.GlobalMassFixer(conf_m)(
                {"y_pred": y.clone(), "x": x.clone()})))["y_pred"]
        for name, t in (("mass", ym), ("water", yw), ("energy", ye), ("chain", yc)):
            assert t.shape == y.shape, (name, t.shape)
            out[f"{tag}_{name}"] = t[0, :, 0].double().numpy()
        print(f"[golden] sigma fixers {tag}: mass dSP max {float((ym - y).abs().max()):.3e}  water dP max "
              f"{float((yw - y).abs().max()):.3e}  energy dT max {float((ye - y).abs().max()):.3e}")
    np.savez_compressed(os.path.join(GOLD, "fixers_sigma.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="pad,T0,T1,glue,swin,swinblock,fuxi,attend,rollC1,rollC3S,rollC3,T0M,layout,fixers,sigma,updown,pre,gen2,rec,asm,gen2loop,fuxi_timm,C1,C3S,C3,T0W,C1W,T0U,T0F,RT,stress")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    for item in args.only.split(","):
        if item == "pad":
            pad_golden()
        elif item == "glue":
            glue_golden()
        elif item == "swin":
            swin_golden()
        elif item == "swinblock":
            swin_block_golden()
        elif item == "fuxi":
            fuxi_golden()
        elif item == "attend":
            attend_golden()
        elif item == "rollC1":      # BASELINE config 2: 24-step rollout on the 1-degree grid
            long_rollout_golden("C1", 24, 20)
        elif item == "rollC3S":     # 8 steps on the 0.25-degree grid (small-width model)
            long_rollout_golden("C3S", 8, 40, with_fp64=False)
        elif item == "rollC1stress":   # 8 steps of the 1-degree model on the "stress" weight family (logits +-40, pre-GELU 1e2)
            long_rollout_golden("C1", 8, 20, family="stress")
        elif item == "rollC3":      # BASELINE config 3 itself: its 40 steps of the FULL-width 124 M-parameter model on the 0.25-degree grid
            long_rollout_golden("C3", 40, 40, with_fp64=False, dense_steps=(1, 10, 20, 40), dense_stride=16)   # (rounds 2-4 stored 6 steps; ~30 min of CPU for 40; round 6: + stride-16 samples of four steps)
        elif item == "layout":
            layout_golden()
        elif item == "fixers":
            fixers_golden()
        elif item == "sigma":
            fixers_sigma_golden()
        elif item == "updown":
            fixers_updown_golden()
        elif item == "pre":
            preblock_golden()
        elif item == "gen2":
            conservation_golden()
        elif item == "rec":
            reconstruct_golden()
        elif item == "asm":
            assemble_golden()
        elif item == "gen2loop":    # the composed gen-2 loop through the reference's own run_forecast
            gen2loop_golden()
        elif item == "fuxi_timm":   # BASELINE config 5's default stage: only where `import timm` works (skips loudly otherwise)
            fuxi_timm_golden()
        elif item in ("T0", "T1", "T0W", "T0U", "T0M", "T0F", "T0H", "T1H", "T0X"):
            model_golden(item, 1, capture_layers=(item in ("T0", "T0W", "T0U")))
        elif item == "RT":   # the model of the reference's own tests/test_crossformer.py
            model_golden(item, 2, False)
        elif item == "C1W":
            model_golden(item, 8, False)
        elif item == "C1":
            model_golden(item, 8, False)
        elif item in ("C3S", "C3"):
            model_golden(item, 16, False)
        elif item == "stressC3S":   # the stress families at HEADLINE map size (721 x 1440): the kernel instantiations only that size selects
            for fam in ("stress", "stress_hi"):
                model_golden("C3S", 16, False, family=fam)
        elif item == "stressC3":    # ... and through the full-width 124 M-parameter model
            for fam in ("stress", "stress_hi"):
                model_golden("C3", 16, False, family=fam)
        elif item == "stress":   # the stress weight families (wxengine.synth.FAMILIES) through the reference: T0 / T1 full maps, C1 strided
            for fam in ("stress", "stress_hi"):
                model_golden("T0", 1, False, family=fam)
                model_golden("T1", 1, False, family=fam)
                model_golden("C1", 8, False, family=fam)
        else:
            raise SystemExit(f"unknown item {item}")


if __name__ == "__main__":
    main()
