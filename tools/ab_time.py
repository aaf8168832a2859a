#!/usr/bin/env python
"""Dev helper (GPU box): same-box A/B of engine switches.  Several engines of one config live in ONE process, each created under its
own environment (the switches are read at wx_create); the arms are timed alternately for several rounds with the benchmark's own loop
(wx_rollout: forward + tracer fixer + de-normalise + next-input assembly), so box-to-box spread (+-3 %) cancels.

    python tools/ab_time.py --config C3 --arm base: --arm one:WX_TWO_STREAM=0 --rounds 3 --steps 20
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "miles-credit_amd"), ROOT]

import torch  # noqa: E402

from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.rollout import channel_layout  # noqa: E402
from wxengine.synth import synth_denorm, synth_forcing, synth_input, synth_state_dict  # noqa: E402


def make(cfg, sd, prec, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng = WXEngine(cfg, prec, 0)
        eng.load_state_dict(sd)
        eng.finalize()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    n_prog, n_static, n_dyn = channel_layout(cfg, n_static=2, n_dyn=2)
    mean, std = synth_denorm(cfg.base_output_channels)
    eng.set_denorm(mean, std)
    eng.set_layout(n_prog, n_static, n_dyn)
    q = list(range(3 * cfg.levels, 4 * cfg.levels))
    eng.set_tracer_fixer(q, [1e-8] * len(q), None, denorm=True)
    return eng, n_dyn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--arm", action="append", default=[], help="name:K=V,K=V (empty list = defaults)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--check-equal", action="store_true", help="assert every arm's final state equals the first arm's bit for bit")
    args = ap.parse_args()
    cfg = named_config(args.config)
    sd = synth_state_dict(cfg)
    arms = []
    for a in args.arm or ["base:"]:
        name, _, kv = a.partition(":")
        env = dict(x.split("=", 1) for x in kv.split(",") if x)
        prec = env.pop("PREC", args.precision)
        eng, n_dyn = make(cfg, sd, prec, env)
        arms.append((name, eng))
    x0 = torch.from_numpy(synth_input(cfg, seed=1000)).cuda()
    frcs = [torch.from_numpy(synth_forcing(cfg, n_dyn, t, seed=1000)).cuda() for t in range(8)]
    oh, ow = cfg.out_hw
    y_phys = torch.empty((1, cfg.base_output_channels, oh, ow), dtype=torch.float32, device="cuda")
    finals = {}
    for name, eng in arms:   # warm-up (and the equality reference)
        xf = torch.empty_like(x0)
        eng.rollout(x0, [frcs[t % 8] for t in range(3)], [y_phys] * 3, x_final=xf)
        torch.cuda.synchronize()
        finals[name] = (xf.clone(), y_phys.clone())
    if args.check_equal:
        ref = finals[arms[0][0]]
        for name, _ in arms[1:]:
            same = torch.equal(finals[name][0], ref[0]) and torch.equal(finals[name][1], ref[1])
            print(f"[ab] {name} vs {arms[0][0]}: {'bit-identical' if same else 'DIFFERENT (max %.3e)' % float((finals[name][1] - ref[1]).abs().max())}")
    res = {name: [] for name, _ in arms}
    for r in range(args.rounds):
        for name, eng in arms:
            xf = torch.empty_like(x0)
            torch.cuda.synchronize()
            t = time.perf_counter()
            eng.rollout(x0, [frcs[t_ % 8] for t_ in range(args.steps)], [y_phys] * args.steps, x_final=xf)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t) / args.steps * 1e3)
    for name, _ in arms:
        v = res[name]
        print(f"[ab] {args.config} {name:>12s}: " + "  ".join(f"{x:.3f}" for x in v) + f"  ms/step  (best {min(v):.3f} = {1e3 / min(v):.1f} steps/s)")


if __name__ == "__main__":
    main()
