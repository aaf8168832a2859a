"""Gen-2 named-tensor forecast loop (wxengine/forecast.py): the routing function against a golden of
the reference's `assemble_rollout_batch`, and -- on the GPU -- three autoregressive steps of the device loop against (a) a golden written
by the reference's OWN `run_forecast` (tests/golden/gen2_loop_T0.npz, tools/make_goldens.py --only gen2loop: reference preblocks, model,
Reconstruct, assemble_rollout_batch, composed by the reference's loop) in every precision and (b) an independent CPU oracle loop."""
import os

import numpy as np
import pytest
import torch

from oracle import preblock_oracle as P
from oracle import wxformer_oracle as O
from wxengine.config import named_config
from wxengine.forecast import InverseScale, assemble_rollout_batch, run_forecast
from wxengine.synth import synth_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden", "assemble_rollout.npz")


def t(v):
    return torch.full((1, 1, 1, 2, 2), float(v))


def test_assemble_rollout_batch_routing_matches_reference():
    g = np.load(GOLD)
    keys = ["era5/prognostic/3d/T", "era5/prognostic/2d/SP", "era5/static/2d/LSM", "era5/dynamic_forcing/2d/tsi",
            "era5/dynamic_forcing/2d/sza", "era5/diagnostic/2d/tp"]
    ic = {"input": {"era5": {k: t(100 + i) for i, k in enumerate(keys)}, "empty": {}}}
    pred = {"era5": {"era5/prognostic/3d/T": t(200), "era5/prognostic/2d/SP": t(201), "era5/diagnostic/2d/tp": t(202)}}
    cur = {"input": {"era5": {"era5/dynamic_forcing/2d/tsi": t(300)}}, "target": None}
    out = assemble_rollout_batch({"y_processed": pred, "ic_preprocessed": ic}, cur, 1)
    assert list(out["input"].keys()) == [str(s) for s in g["sources"]]
    assert list(out["input"]["era5"].keys()) == [str(k) for k in g["keys"]]
    assert [float(v.flatten()[0]) for v in out["input"]["era5"].values()] == list(g["vals"])
    assert out["input"]["era5"]["era5/prognostic/3d/T"] is pred["era5"]["era5/prognostic/3d/T"]   # routed, not copied
    with pytest.raises(TypeError):
        assemble_rollout_batch({"y_processed": torch.zeros(1), "ic_preprocessed": ic}, cur)


def schema(cfg):
    L = cfg.levels
    inp = [(f"era5/prognostic/3d/{v}", L) for v in "UVTQ"] + [(f"era5/prognostic/2d/s{i}", 1) for i in range(cfg.surface_channels)]
    inp += [("era5/static/2d/LSM", 1), ("era5/static/2d/Z", 1), ("era5/dynamic_forcing/2d/tsi", 1), ("era5/dynamic_forcing/2d/sza", 1)]
    out = inp[:4 + cfg.surface_channels] + [(f"era5/diagnostic/2d/d{i}", 1) for i in range(cfg.output_only_channels)]
    return inp, out


@pytest.mark.gpu
def test_three_step_device_loop_vs_oracle_loop():
    from wxengine.model import WXFormerHIP
    cfg = named_config("T0")
    sd = synth_state_dict(cfg)
    inp, out = schema(cfg)
    gen = np.random.Generator(np.random.Philox(key=[77, 1]))
    H, W = cfg.image_height, cfg.image_width

    def field(nl, scale=1.0, shift=0.0):
        return torch.from_numpy((gen.standard_normal((1, nl, 1, H, W)) * scale + shift).astype(np.float32))
    mean = {k.split("/")[-1]: (np.arange(nl, dtype=np.float32) * 0.1 + 0.3) for k, nl in inp[:-2]}
    std = {k.split("/")[-1]: (np.arange(nl, dtype=np.float32) * 0.2 + 1.5) for k, nl in inp[:-2]}
    mean.update({f"d{i}": np.float32(0.1 * i) for i in range(cfg.output_only_channels)})
    std.update({f"d{i}": np.float32(2.0 + i) for i in range(cfg.output_only_channels)})
    ic = {"input": {"era5": {k: field(nl, 1.5, 0.3) for k, nl in inp}}}
    frcs = [{"input": {"era5": {k: field(1) for k, _ in inp[-2:]}}} for _ in range(2)]
    cmap, cur = {}, 0
    for k, nl in out:
        cmap[k] = {"slice": slice(cur, cur + nl), "orig_shape": (nl, 1)}
        cur += nl
    # ---- oracle loop (CPU)
    want = []
    x_named = ic["input"]
    for step in range(3):
        x, _ = P.assemble(x_named, mean, std)
        y = O.forward(cfg, sd, x.numpy())                      # [1, C_out, 1, H, W]
        named = {k: y[:, v["slice"]] for k, v in cmap.items()}
        for k in named:
            n = k.split("/")[-1]
            named[k] = named[k] * torch.as_tensor(std[n]).reshape(1, -1, 1, 1, 1) + torch.as_tensor(mean[n]).reshape(1, -1, 1, 1, 1)
        want.append({k: v.clone() for k, v in named.items()})
        if step < 2:
            nxt = {}
            for k, _ in inp:
                ft = k.split("/")[1]
                nxt[k] = named[k] if ft == "prognostic" else (frcs[step]["input"]["era5"][k] if ft == "dynamic_forcing" else ic["input"]["era5"][k])
            x_named = {"era5": nxt}
    # ---- device loop
    mc = dict(image_height=37, image_width=72, frames=1, channels=4, surface_channels=4, input_only_channels=4,
              output_only_channels=3, levels=3, dim=[32, 64, 128, 256], depth=[1, 1, 2, 1],
              global_window_size=[4, 2, 2, 1], local_window_size=3,
              cross_embed_kernel_sizes=[[4, 8, 16, 32], [2, 4], [2, 4], [2, 4]], cross_embed_strides=[2, 2, 2, 2],
              padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]), post_conf=dict(activate=False))
    model = WXFormerHIP(precision="fp32", **mc).to("cuda").eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cu = lambda b: {"input": {s: {k: v.cuda() for k, v in d.items()} for s, d in b["input"].items()}}  # noqa: E731
    got = []
    run_forecast(model, cu(ic), [cu(f) for f in frcs], 3, cmap, mean, std, [InverseScale(mean, std)],
                 lambda yp, step: got.append({k: v.cpu() for k, v in yp["era5"].items()}))
    assert len(got) == 3
    for step in range(3):
        for k in want[step]:
            a, b = got[step][k].numpy(), want[step][k].numpy()
            assert a.shape == b.shape
            assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1.0), (step, k)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "fp32s", "bf16"])
def test_three_step_device_loop_vs_reference_run_forecast(prec):
    """The composed loop against the reference's own: same IC, forcings and statistics (tests/synth_batches.gen2loop_batches), three steps
    of `credit.trainers.rollout_utils.run_forecast` on the reference CrossFormer stored by tools/make_goldens.py --only gen2loop."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from synth_batches import gen2loop_batches, gen2loop_schema
    from wxengine.model import WXFormerHIP
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gen2_loop_T0.npz"))
    cfg = named_config("T0")
    sd = synth_state_dict(cfg)
    ic, frcs, mean, std = gen2loop_batches(cfg, 3)
    _, out = gen2loop_schema(cfg)
    cmap, cur = {}, 0
    for k, nl in out:
        cmap[k] = {"slice": slice(cur, cur + nl), "orig_shape": (nl, 1)}
        cur += nl
    mc = dict(image_height=37, image_width=72, frames=1, channels=4, surface_channels=4, input_only_channels=4,
              output_only_channels=3, levels=3, dim=[32, 64, 128, 256], depth=[1, 1, 2, 1],
              global_window_size=[4, 2, 2, 1], local_window_size=3,
              cross_embed_kernel_sizes=[[4, 8, 16, 32], [2, 4], [2, 4], [2, 4]], cross_embed_strides=[2, 2, 2, 2],
              padding_conf=dict(activate=True, mode="earth", pad_lat=[6, 6], pad_lon=[12, 12]), post_conf=dict(activate=False))
    model = WXFormerHIP(precision=prec, **mc).to("cuda").eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cu = lambda b: {"input": {s: {k: v.cuda() for k, v in d.items()} for s, d in b["input"].items()}}  # noqa: E731
    got = []
    run_forecast(model, cu(ic), [cu(f) for f in frcs], 3, cmap, mean, std, [InverseScale(mean, std)],
                 lambda yp, step: got.append({k: v.cpu() for k, v in yp["era5"].items()}))
    assert len(got) == 3 and list(got[0].keys()) == [str(k) for k in g["keys"]]
    worst = 0.0
    for step in range(3):
        for k in got[step]:
            a, b = got[step][k].numpy()[..., ::2, ::2], g[f"step{step}:{k}"]
            assert a.shape == b.shape, (step, k, a.shape, b.shape)
            if prec in ("fp32", "fp32s"):
                err = np.abs(a - b).max() / max(np.abs(b).max(), 1.0)
                assert err <= 2e-4, (prec, step, k, err)
            else:
                err = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
                assert err <= 3e-2, (prec, step, k, err)
            worst = max(worst, float(err))
    print(f"[gen-2 loop vs reference run_forecast] {prec}: worst {'rel max' if prec != 'bf16' else 'rel-L2'} error over 3 steps x {len(got[0])} variables {worst:.2e}")
