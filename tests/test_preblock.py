"""Input side of the step: the preblock oracle against the reference's ERA5Normalizer + ConcatToTensor
(tests/golden/preblock.npz), the host ordering logic, and -- on the GPU -- the fused device kernel through the C ABI."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]

from oracle import preblock_oracle as P  # noqa: E402
from wxengine import preblock as host  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "preblock.npz")


def batch():
    from synth_batches import preblock_batch
    return preblock_batch()


def test_oracle_matches_reference_golden():
    g = np.load(GOLD)
    b, mean, std = batch()
    x, cmap = P.assemble(b["input"], mean, std)
    np.testing.assert_array_equal(x.numpy(), g["x"])            # same fp32 operations in the same order: bit exact
    assert list(cmap.keys()) == [str(k) for k in g["keys"]]
    assert [v["slice"].start for v in cmap.values()] == list(g["starts"])
    assert [v["slice"].stop for v in cmap.values()] == list(g["stops"])
    x_raw, _ = P.assemble(b["input"])
    np.testing.assert_array_equal(x_raw.numpy(), g["x_raw"])


def test_host_order_and_stats():
    g = np.load(GOLD)
    b, mean, std = batch()
    keys = host.ordered_keys(b["input"])
    assert keys == [str(k) for k in g["keys"]]
    levels = [b["input"]["era5"][k].shape[1] for k in keys]
    m, s = host.channel_stats(keys, levels, mean, std)
    assert m.shape == (17,) and s.shape == (17,)
    assert m[0] == 210.0 and s[6] == 0.0          # the zero std reaches the device as is; the kernel clamps (norm.py:98)
    assert m[14] == 0.0 and s[14] == 1.0          # LSM has no statistics: passes through
    with pytest.raises(ValueError):
        host.channel_stats(keys, levels, {"T": np.zeros(3)}, {"T": np.ones(3)})


@pytest.mark.gpu
def test_device_preblock_bit_exact():
    g = np.load(GOLD)
    b, mean, std = batch()
    pre = host.DevicePreblock(b["input"], mean, std)
    assert pre.channels == 17 and list(pre.channel_map.keys()) == [str(k) for k in g["keys"]]
    x = pre({"era5": {k: v.cuda() for k, v in b["input"]["era5"].items()}})
    torch.cuda.synchronize()
    np.testing.assert_array_equal(x.cpu().numpy(), g["x"])      # (t - mean) / max(std, 1e-12) in fp32: bit exact
    raw = host.DevicePreblock(b["input"])
    np.testing.assert_array_equal(raw(b["input"]).cpu().numpy(), g["x_raw"])
    # full-size grid, odd width (scalar tail path): against the oracle
    gen = np.random.Generator(np.random.Philox(key=[5, 5]))
    big = {"era5": {"era5/prognostic/3d/T": torch.from_numpy(gen.standard_normal((1, 3, 2, 181, 359)).astype(np.float32)),
                    "era5/prognostic/2d/SP": torch.from_numpy(gen.standard_normal((1, 1, 2, 181, 359)).astype(np.float32))}}
    st_m, st_s = {"T": np.array([1., 2., 3.], np.float32), "SP": np.float32(0.5)}, {"T": np.array([2., 4., 0.], np.float32), "SP": np.float32(3.)}
    ref, _ = P.assemble(big, st_m, st_s)
    got = host.DevicePreblock(big, st_m, st_s)(big)
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())
