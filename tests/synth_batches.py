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
