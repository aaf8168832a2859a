#!/usr/bin/env python
"""What a weight family of wxengine.synth actually does to the REFERENCE model (dev container only: imports /root/reference).

Hooks the reference CrossFormer and prints, per stage: |mean| / sigma of every channel-LayerNorm input row (median / max over pixels),
the largest |logit| entering a softmax, the largest FeedForward pre-GELU magnitude, and |mean| / sigma of the GroupNorm groups.
Used to calibrate the "stress" / "stress_hi" families -- the numbers it prints are quoted in DESIGN.md.

    python tools/stress_stats.py T0 stress
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "miles-credit_amd"), ROOT]

import oracle_stub  # noqa: E402

oracle_stub.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from make_goldens import reference_model  # noqa: E402
from wxengine.config import named_config  # noqa: E402
from wxengine.synth import synth_input  # noqa: E402


def main():
    name, family = sys.argv[1], sys.argv[2]
    cfg = named_config(name)
    m = reference_model(cfg, family=family)
    from credit.models.crossformer import LayerNorm
    rows = []

    def ln_hook(n):
        def f(_m, inp):
            x = inp[0]
            mu = x.mean(dim=1)
            sd = x.var(dim=1, unbiased=False).sqrt()
            r = (mu.abs() / sd.clamp_min(1e-30)).flatten()
            rows.append((n, "LN |mean|/sigma", float(r.median()), float(r.max())))
        return f

    def gelu_hook(n):
        def f(_m, inp):
            rows.append((n, "pre-GELU |x|", float(inp[0].abs().median()), float(inp[0].abs().max())))
        return f

    def gn_hook(n):
        def f(mod, inp):
            x = inp[0]
            b, c, h, w = x.shape
            g = x.reshape(b, mod.num_groups, -1)
            r = (g.mean(dim=2).abs() / g.var(dim=2, unbiased=False).sqrt().clamp_min(1e-30)).flatten()
            rows.append((n, "GN |mean|/sigma", float(r.median()), float(r.max())))
        return f

    for n, mod in m.named_modules():
        if isinstance(mod, LayerNorm):
            mod.register_forward_pre_hook(ln_hook(n))
        elif isinstance(mod, torch.nn.GELU):
            mod.register_forward_pre_hook(gelu_hook(n))
        elif isinstance(mod, torch.nn.GroupNorm):
            mod.register_forward_pre_hook(gn_hook(n))
    logits = []
    orig = torch.Tensor.softmax

    def spy(self, *a, **k):
        logits.append(float(self.abs().max()))
        return orig(self, *a, **k)

    torch.Tensor.softmax = spy
    try:
        with torch.no_grad():
            y = m(torch.from_numpy(synth_input(cfg)))
    finally:
        torch.Tensor.softmax = orig
    for r in rows:
        print(f"{r[0]:42s} {r[1]:18s} median {r[2]:10.3f}  max {r[3]:12.3f}")
    print("softmax |logit| max per attention:", " ".join(f"{v:.1f}" for v in logits))
    print(f"y: mean|y| {float(y.abs().mean()):.4f}  max|y| {float(y.abs().max()):.4f}  finite {bool(torch.isfinite(y).all())}")


if __name__ == "__main__":
    main()
