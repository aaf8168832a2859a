"""Synthetic named-tensor batches shared by tools/make_goldens.py (golden generation) and the tests (pure numpy / torch:
importable on the GPU box, where the reference is absent)."""
import numpy as np
import torch


def preblock_batch(seed=21):
    """A synthetic gen-2 style batch: deliberately NOT in canonical order, with per-level, scalar and missing statistics."""
    g = np.random.Generator(np.random.Philox(key=[seed, 1]))
    B, T, H, W = 2, 1, 9, 14
    def f(nl):
        return torch.from_numpy(g.standard_normal((B, nl, T, H, W)).astype(np.float32) * 7.0 + 3.0)
    variables = {"era5/dynamic_forcing/2d/tsi": f(1), "era5/prognostic/2d/SP": f(1), "era5/static/2d/LSM": f(1),
                 "era5/prognostic/3d/T": f(4), "era5/prognostic/3d/Q": f(4), "era5/prognostic/2d/t2m": f(1),
                 "era5/static/2d/Z_GDS4_SFC": f(1), "era5/prognostic/3d/U": f(4)}
    mean = {"T": np.array([210., 230., 260., 280.], np.float32), "Q": np.array([1e-6, 1e-4, 2e-3, 8e-3], np.float32),
            "U": np.array([5., 3., 1., 0.], np.float32), "SP": np.float32(9.8e4), "t2m": np.float32(285.), "tsi": np.float32(1.2e6)}
    std = {"T": np.array([8., 9., 12., 15.], np.float32), "Q": np.array([1e-6, 2e-4, 0.0, 5e-3], np.float32),   # a zero std: clamp path
           "U": np.array([20., 15., 10., 6.], np.float32), "SP": np.float32(9.0e3), "t2m": np.float32(15.), "tsi": np.float32(9.0e5)}
    return {"input": {"era5": variables}}, mean, std


def conservation_batch(seed=31, midpoint=True):
    """A gen-2 style `batch_dict` on the reference's 10 x 18 demo grid with 7 hybrid levels (6 mid-level values when
    `midpoint`): y_processed (t1) and x_physical (t0, two frames), physical units, B = 2."""
    g = np.random.Generator(np.random.Philox(key=[seed, 2]))
    B, H, W, L = 2, 10, 18, 6 if midpoint else 7

    def t(a):
        return torch.from_numpy(np.asarray(a, np.float32))

    def state(T):
        return {"cam/prognostic/3d/T": t(250.0 + 30.0 * g.standard_normal((B, L, T, H, W))),
                "cam/prognostic/3d/Qtot": t(np.abs(0.004 + 0.004 * g.standard_normal((B, L, T, H, W)))),
                "cam/prognostic/3d/U": t(12.0 * g.standard_normal((B, L, T, H, W))),
                "cam/prognostic/3d/V": t(8.0 * g.standard_normal((B, L, T, H, W))),
                "cam/prognostic/2d/PS": t(1.0e5 + 2.0e3 * g.standard_normal((B, 1, T, H, W)))}
    y = state(1)
    for k in ("FSUTOA", "FLUT", "FSDS", "FSUS", "FLDS", "FLUS", "SHFLX", "LHFLX"):
        scale = 200.0 if k in ("FSUTOA", "FLUT") else 3.0e6        # TOA terms in W/m2, surface terms in J/m2 (conservation.py)
        y[f"cam/diagnostic/2d/{k}"] = t(np.abs(scale * g.standard_normal((B, 1, 1, H, W))))
    y["cam/diagnostic/2d/PRECT"] = t(np.abs(2e-3 * g.standard_normal((B, 1, 1, H, W))))
    y["cam/diagnostic/2d/QFLX"] = t(-np.abs(1e-3 * g.standard_normal((B, 1, 1, H, W))))
    x = state(2)
    x["cam/dynamic_forcing/2d/SOLIN"] = t(np.abs(300.0 * g.standard_normal((B, 1, 2, H, W))))
    gph = (50.0 + 20.0 * g.standard_normal((H, W))).astype(np.float32)
    return {"y_processed": {"cam": y}, "x_physical": {"cam": x}}, gph


def odd_config(h3, w3, lw, gw, pad=None, depth=(1, 1, 1, 1), arch="crossformer"):
    """Small CrossFormer geometries outside the BASELINE family (stage-3 map h3 x w3, local window lw, long windows gw, optional
    asymmetric earth padding ((top, bottom), (left, right))): used to sweep the lat-band plan and the engine over window / rank
    combinations the named configs do not hit."""
    from wxengine.config import WXConfig
    mc = dict(frames=1, channels=2, surface_channels=2, input_only_channels=2, output_only_channels=1, levels=2,
              image_height=16 * h3 - (sum(pad[0]) if pad else 0), image_width=16 * w3 - (sum(pad[1]) if pad else 0),
              patch_width=1, patch_height=1, cross_embed_kernel_sizes=[[4, 8, 16, 32], [2, 4], [2, 4], [2, 4]],
              cross_embed_strides=[2, 2, 2, 2], dim=[32, 64, 128, 256], depth=list(depth), global_window_size=list(gw),
              local_window_size=lw, interp=True, use_spectral_norm=True,
              padding_conf=dict(activate=bool(pad), mode="earth", pad_lat=list(pad[0]) if pad else [0, 0],
                                pad_lon=list(pad[1]) if pad else [0, 0]))
    return WXConfig.from_model_conf(mc, arch=arch)


ODD_CONFIGS = {
    "w2": dict(h3=2, w3=4, lw=2, gw=(4, 4, 2, 2)),                         # a long window at the deepest stage too
    "w3": dict(h3=3, w3=3, lw=3, gw=(8, 4, 2, 1)),
    "w1": dict(h3=4, w3=4, lw=1, gw=(2, 2, 2, 2)),                         # 1-token local windows
    "w16": dict(h3=6, w3=6, lw=2, gw=(16, 8, 4, 2), depth=(1, 1, 2, 1)),   # 256-token long windows
    "w5p": dict(h3=5, w3=5, lw=5, gw=(8, 4, 2, 1), pad=((13, 11), (9, 7))),  # asymmetric pads, odd image
}


def two_source_conf():
    """A CREDIT config whose model input interleaves two data sources (channel_utils.py:161-250: a field type's channels are
    contiguous only within a source).  x: 14 channels, y: 11, forcing tensor: 3."""
    def grp(v3=(), v2=()):
        return {"vars_3D": list(v3), "vars_2D": list(v2)}
    return {"model": {"levels": 3},
            "data": {"history_len": 1, "source": {
                "era5": {"levels": [500, 700, 850], "variables": {
                    "prognostic": grp(("U", "T"), ("SP",)), "static": grp(v2=("LSM",)),
                    "dynamic_forcing": grp(v2=("tsi",)), "diagnostic": grp(v2=("precip",))}},
                "aux": {"levels": None, "variables": {
                    "prognostic": grp(v2=("sst", "ice")), "static": grp(v2=("depth",)),
                    "dynamic_forcing": grp(v2=("tide", "wind")), "diagnostic": grp(v2=("flux",))}}}}}


def gen2loop_schema(cfg):
    """Variable keys of a gen-2 forecast on a CrossFormer config: (input keys with level counts, output keys with level counts)."""
    L = cfg.levels
    inp = [(f"era5/prognostic/3d/{v}", L) for v in "UVTQ"[:cfg.channels]] + [(f"era5/prognostic/2d/s{i}", 1) for i in range(cfg.surface_channels)]
    inp += [("era5/static/2d/LSM", 1), ("era5/static/2d/Z", 1), ("era5/dynamic_forcing/2d/tsi", 1), ("era5/dynamic_forcing/2d/sza", 1)]
    out = inp[:cfg.channels + cfg.surface_channels] + [(f"era5/diagnostic/2d/d{i}", 1) for i in range(cfg.output_only_channels)]
    return inp, out


def gen2loop_batches(cfg, n_steps=3, seed=77):
    """Initial condition, n_steps - 1 forcing batches (physical units, [1, n_levels, 1, H, W]) and per-variable statistics for the
    composed gen-2 loop (tools/make_goldens.py --only gen2loop drives the reference's run_forecast with exactly these)."""
    inp, out = gen2loop_schema(cfg)
    gen = np.random.Generator(np.random.Philox(key=[seed, 1]))
    H, W = cfg.image_height, cfg.image_width

    def field(nl, scale=1.0, shift=0.0):
        return torch.from_numpy((gen.standard_normal((1, nl, 1, H, W)) * scale + shift).astype(np.float32))
    mean = {k.split("/")[-1]: (np.arange(nl, dtype=np.float32) * 0.1 + 0.3) for k, nl in inp[:-2]}
    std = {k.split("/")[-1]: (np.arange(nl, dtype=np.float32) * 0.2 + 1.5) for k, nl in inp[:-2]}
    mean.update({f"d{i}": np.float32(0.1 * i) for i in range(cfg.output_only_channels)})
    std.update({f"d{i}": np.float32(2.0 + i) for i in range(cfg.output_only_channels)})
    ic = {"input": {"era5": {k: field(nl, 1.5, 0.3) for k, nl in inp}}}
    frcs = [{"input": {"era5": {k: field(1) for k, _ in inp[-2:]}}} for _ in range(n_steps - 1)]
    return ic, frcs, mean, std
