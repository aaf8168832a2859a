"""Gen-2 name-keyed conservation fixers (wxengine/conservation.py over the device PostBlock) against a golden of the
reference's credit/postblock/conservation.py chain (tracer -> mass -> water -> energy up/down) on a hybrid sigma grid."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
from synth_batches import conservation_batch  # noqa: E402
from wxengine import conservation as G2  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "conservation_gen2.npz")
SIGMA = os.path.join(os.path.dirname(__file__), "golden", "fixers_sigma.npz")


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("midpoint", [True, False])
def test_gen2_chain_matches_reference(midpoint):
    g = np.load(GOLD)
    coef = np.load(SIGMA)
    tag = "mid" if midpoint else "trapz"
    batch, gph = conservation_batch(midpoint=midpoint)
    batch = {k: {s: {n: t.cuda() for n, t in v.items()} for s, v in d.items()} for k, d in batch.items()}
    lat = np.array([90, 70, 50, 30, 10, -10, -30, -50, -70, -90], dtype=np.float32)
    lon2d, lat2d = np.meshgrid(np.arange(0, 360, 20, dtype=np.float32), lat)
    phys = dict(lat2d=lat2d, lon2d=lon2d, coef_a=coef["coef_a"], coef_b=coef["coef_b"], grid_type="sigma", midpoint=midpoint,
                gph_surf=gph)
    P, D = "cam/prognostic/", "cam/diagnostic/2d/"
    chain = [G2.TracerFixer([P + "3d/Qtot", D + "PRECT"], [1e-9, 0.0], [None, 1.0]),
             G2.GlobalMassFixer(P + "3d/Qtot", P + "2d/PS", **phys),
             G2.GlobalWaterFixer(P + "3d/Qtot", P + "2d/PS", D + "PRECT", D + "QFLX", 6, **phys),
             G2.GlobalEnergyFixerUpDown(P + "3d/T", P + "3d/Qtot", P + "3d/U", P + "3d/V", P + "2d/PS", ["PHIS"],
                                        "cam/dynamic_forcing/2d/SOLIN", D + "FSUTOA", D + "FLUT", D + "FSDS", D + "FSUS", D + "FLDS",
                                        D + "FLUS", D + "SHFLX", D + "LHFLX", 6, **phys)]
    for f in chain:
        batch = f(batch)
    torch.cuda.synchronize()
    got = {k: batch["y_processed"]["cam"][k].cpu().numpy() for k in (P + "2d/PS", D + "PRECT", P + "3d/T", P + "3d/Qtot")}
    for k, v in got.items():
        assert v.shape == g[f"{tag}:{k}"].shape, k
    assert rel(got[P + "2d/PS"], g[f"{tag}:{P}2d/PS"]) < 1e-5
    assert rel(got[P + "3d/T"], g[f"{tag}:{P}3d/T"]) < 1e-4
    assert rel(got[D + "PRECT"], g[f"{tag}:{D}PRECT"]) < 2e-3     # ratio of a small difference of large global integrals
    np.testing.assert_array_equal(got[P + "3d/Qtot"], g[f"{tag}:{P}3d/Qtot"])   # clamp only: bit exact
