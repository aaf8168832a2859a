"""CPU restatements of small device functions whose constants matter, evaluated in the device's precision (fp32) against fp64 truth.
The constants are parsed out of the shipped header, so an edit there is what gets tested."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "miles-credit_amd", "csrc")


def _gelu_as_constants():
    src = open(os.path.join(CSRC, "wx_common.h")).read()
    body = src[src.index("__device__ inline float gelu_as(float x) {"):]
    body = body[:body.index("\n}\n")]
    num = r"(-?[0-9]+\.[0-9]+)f"
    p, inv_sqrt2 = map(float, re.search(r"fmaf\(ax, " + num + r" \* " + num + r", 1\.0f\)", body).groups())
    e = float(re.search(r"x \* x \* " + num, body).group(1))
    coef = [float(c) for c in re.findall(r"0\.5f \* " + num, body)]
    clamp = float(re.search(r"fminf\(ax, ([0-9.]+)f\)", body).group(1))
    assert len(coef) == 5
    return p * inv_sqrt2, e, coef, clamp


def test_gelu_as_host_model():
    """wx_common.h gelu_as (the split-bf16 FeedForward's activation: Abramowitz & Stegun 7.1.26 on rcp / exp2): in fp32 evaluation
    within 5e-7 absolute of the exact GELU over [-12, 12] -- the libm erff form's own error is 4.5e-7 -- and exact limits outside
    (x for large x, -0 for very negative x)."""
    pz, e2, (a5, a4, a3, a2, a1), clamp = _gelu_as_constants()
    f = np.float32
    x = np.linspace(-12, 12, 1_200_001).astype(f)
    ax = np.abs(x)
    t = (f(1) / (ax * f(pz) + f(1))).astype(f)
    e = np.exp2((x * x * f(e2)).astype(f)).astype(f)
    pl = (t * f(0.5 * a5) + f(0.5 * a4)).astype(f)
    for c in (a3, a2, a1):
        pl = (pl * t + f(0.5 * c)).astype(f)
    pl = (pl * t).astype(f)
    y = (np.maximum(x, f(0)) - np.minimum(ax, f(clamp)) * (pl * e).astype(f)).astype(f)
    x64 = x.astype(np.float64)
    ref = 0.5 * x64 * (1.0 + erf(x64 / np.sqrt(2.0)))
    assert np.abs(y - ref).max() <= 5e-7
    erff_form = (f(0.5) * x * (f(1) + erf((x * f(0.70710678)).astype(f)).astype(f))).astype(f)
    assert np.abs(y - ref).max() <= 1.25 * np.abs(erff_form - ref).max()
    assert abs(e2 + 0.5 * np.log2(np.e)) < 1e-9 and abs(pz - 0.3275911 / np.sqrt(2.0)) < 1e-7
    big = np.array([20.0, 1e4, 3e38], f)
    assert np.array_equal(np.maximum(big, f(0)) - np.minimum(big, f(clamp)) * f(0), big)
