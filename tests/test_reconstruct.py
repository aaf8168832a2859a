"""Gen-2 Reconstruct / FlattenToTensor mirrors (wxengine/reconstruct.py) against a golden of the reference's
credit/postblock/reconstruct.py."""
import os

import numpy as np
import pytest
import torch

from wxengine.reconstruct import FlattenToTensor, Reconstruct

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reconstruct.npz")
CMAP = {"era5/prognostic/3d/T": {"slice": slice(0, 4), "orig_shape": (4, 1)},
        "era5/prognostic/3d/Q": {"slice": slice(4, 8), "orig_shape": (4, 1)},
        "era5/prognostic/2d/SP": {"slice": slice(8, 9), "orig_shape": (1, 1)},
        "era5/diagnostic/2d/tp": {"slice": slice(9, 10), "orig_shape": (1, 1)},
        "era5/diagnostic/2d/evap": {"slice": slice(10, 11), "orig_shape": (1, 1)}}


def test_reconstruct_views_match_reference():
    g = np.load(GOLD)
    y = torch.from_numpy(g["y"])
    bd = Reconstruct()({"y_pred": y, "metadata": {"target": {"_channel_map": CMAP}}})
    for k in CMAP:
        v = bd["y_processed"]["era5"][k]
        np.testing.assert_array_equal(v.numpy(), g["rec:" + k])
        assert v.data_ptr() >= y.data_ptr() and v.data_ptr() < y.data_ptr() + y.numel() * 4   # a view, not a copy


@pytest.mark.gpu
def test_flatten_on_device_matches_reference():
    g = np.load(GOLD)
    y = torch.from_numpy(g["y"]).cuda()
    bd = Reconstruct()({"y_pred": y, "metadata": {"target": {"_channel_map": CMAP}}})
    bd["y_processed"]["era5"]["era5/prognostic/2d/SP"] = bd["y_processed"]["era5"]["era5/prognostic/2d/SP"] * 2.0
    bd = FlattenToTensor()(bd)
    np.testing.assert_array_equal(bd["y_pred"].cpu().numpy(), g["flat"])
    # with a forward scaler: (t - mean) / std per variable, against plain torch
    mean, std = {"T": np.array([1., 2., 3., 4.], np.float32), "SP": np.float32(5.0)}, {"T": np.array([2., 2., 4., 4.], np.float32), "SP": np.float32(0.5)}
    bd2 = FlattenToTensor(mean, std)(bd)
    ref = torch.from_numpy(g["flat"]).clone()
    ref[:, 0:4] = (ref[:, 0:4] - torch.tensor(mean["T"]).view(1, 4, 1, 1)) / torch.tensor(std["T"]).view(1, 4, 1, 1)
    ref[:, 8:9] = (ref[:, 8:9] - 5.0) / 0.5
    np.testing.assert_array_equal(bd2["y_pred"].cpu().numpy(), ref.numpy())
