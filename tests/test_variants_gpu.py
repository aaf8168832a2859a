"""The launch-shape variants the engine picks by itself (split-K, the CrossEmbed chunk split, the merged ConvTranspose parity launch) against
the same engine with the variant switched off: same model, same input, two engines in one process.

* merged parity convs: every output element is the same K walk in the same order -> bit-identical;
* split-K / chunk split: fp32 partial sums are added in a different (fixed) order -> equal to fp32 rounding in the fp32 engine, inside
  bf16 rounding in the bf16 engine; each variant is deterministic (two runs bit-identical).
C1 (BASELINE config 2, the 1-degree grid) is the configuration where all three trigger.
"""
import os

import numpy as np
import pytest
import torch

from wxengine.config import named_config
from wxengine.engine import WXEngine
from wxengine.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu


def _engine(name, prec, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cfg = named_config(name)
        eng = WXEngine(cfg, prec, 0)   # the switches are read when the engine is created
        eng.load_state_dict(synth_state_dict(cfg))
        eng.finalize()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return eng


def _forward(eng, x):
    return eng.forward(x).clone()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_split_variants_match_unsplit(prec):
    cfg = named_config("C1")
    x = torch.from_numpy(synth_input(cfg)).cuda()
    on = _engine("C1", prec, {})
    off = _engine("C1", prec, {"WX_NO_SPLIT_K": "1", "WX_NO_EMBED_SPLIT": "1"})
    y_on, y_on2, y_off = _forward(on, x), _forward(on, x), _forward(off, x)
    assert torch.equal(y_on, y_on2), "the split launches must be deterministic (fixed summation order)"
    scale = float(y_off.abs().max())
    err = float((y_on - y_off).abs().max())
    if prec == "fp32":
        assert err <= 2e-5 * scale, f"split vs unsplit (fp32): {err:.3e} of {scale:.3e}"
    else:
        l2 = float(torch.linalg.norm((y_on - y_off).double()) / torch.linalg.norm(y_off.double()))
        assert l2 <= 1e-2, f"split vs unsplit (bf16) rel-L2 {l2:.3e}"
    assert not torch.equal(y_on, y_off) or prec == "fp32", "the variants should actually differ in summation order (is the split taken?)"


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_merged_parity_convs_bit_identical(prec):
    cfg = named_config("C1")   # type: crossformer -> ConvTranspose k4 s2 p1 as the last up-block
    x = torch.from_numpy(synth_input(cfg)).cuda()
    merged = _engine("C1", prec, {})
    separate = _engine("C1", prec, {"WX_NO_MERGE_PARITY": "1"})
    assert torch.equal(_forward(merged, x), _forward(separate, x))
