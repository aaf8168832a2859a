"""Dev-container-only pin: the oracle against the imported reference itself (skipped elsewhere)."""
import numpy as np
import pytest
import torch

from oracle import wxformer_oracle as O
from wxengine.config import named_config
from wxengine.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def ref():
    import oracle_stub
    oracle_stub.install()
    import make_goldens
    return make_goldens


@pytest.mark.parametrize("name", ["T0", "T1"])
def test_forward_and_every_block(ref, name):
    cfg = named_config(name)
    m = ref.reference_model(cfg)
    sd = synth_state_dict(cfg)
    x = torch.from_numpy(synth_input(cfg))
    caps = {}
    hooks = [mod.register_forward_hook(lambda _m, _i, o, n=n: caps.__setitem__(n, o.detach()))
             for n, mod in m.named_modules() if n.count(".") <= 4 and n and not n.startswith("cube")]
    with torch.no_grad():
        yr = m(x)
    for h in hooks:
        h.remove()
    mine = {}
    yo = O.forward(cfg, sd, x, capture=mine)
    assert float((yr - yo).abs().max()) <= 2e-5 * float(yr.abs().max())
    checked = 0
    for k, v in mine.items():
        if k in caps and caps[k].shape == v.shape:
            assert float((caps[k] - v).abs().max()) <= 2e-5 * max(1.0, float(v.abs().max())), k
            checked += 1
    assert checked >= 8


def test_dpb_bias_matches_reference_attention(ref):
    cfg = named_config("T1")
    m = ref.reference_model(cfg)
    sd = synth_state_dict(cfg)
    att = m.layers[0][1].layers[0][0]
    wsz = att.window_size
    pos = torch.arange(-wsz, wsz + 1)
    rel = torch.stack(torch.meshgrid(pos, pos, indexing="ij")).reshape(2, -1).t().float()
    with torch.no_grad():
        want = att.dpb(rel)[att.rel_pos_indices]
    got = O.dpb_bias(sd, "layers.0.1.layers.0.0.dpb", wsz)
    assert float((want - got).abs().max()) < 1e-5


def test_fp64_oracle_bounds_fp32_noise(ref):
    cfg = named_config("T0")
    sd = synth_state_dict(cfg)
    x = synth_input(cfg)
    y32 = O.forward(cfg, sd, x)
    y64 = O.forward(cfg, sd, x, dtype=torch.float64)
    assert float((y32 - y64).abs().max()) < 1e-5
