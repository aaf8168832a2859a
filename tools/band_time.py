"""GPU box: what lat-band sharding costs and buys, measured with VIRTUAL ranks on one GPU.
For n ranks: per-rank kernel time of a step (HIP events around every launch, incl. the exchange pack / unpack kernels),
the slowest rank (= the critical path a real n-GPU run cannot beat), bytes each rank sends per step, parity vs unsharded.
    python tools/band_time.py C3 bf16 2 4 8"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "miles-credit_amd"))
from wxengine.config import named_config  # noqa: E402
from wxengine.engine import WXEngine  # noqa: E402
from wxengine.latband import VirtualBands  # noqa: E402
from wxengine.synth import synth_input, synth_state_dict  # noqa: E402

name, prec = sys.argv[1], sys.argv[2]
cfg = named_config(name)
sd = synth_state_dict(cfg)
x = torch.from_numpy(synth_input(cfg)).cuda()
ref = WXEngine(cfg, prec, 0)
ref.load_state_dict(sd)
ref.finalize()
ref.profile(1)
ref.step(x, None, want_phys=False, want_next=False)
ref.profile_reset()
K = 3
for _ in range(K):
    y0, _, _ = ref.step(x, None, want_phys=False, want_next=False)
torch.cuda.synchronize()
t_ref = sum(k["ms"] for k in ref.profile_read()) / K
print(f"{name} {prec} unsharded: {t_ref:.3f} ms kernel time / step")
out = {"config": name, "precision": prec, "unsharded_ms": t_ref, "bands": []}
del ref
for n in [int(a) for a in sys.argv[3:]]:
    vb = VirtualBands(cfg, sd, n, prec)
    for r in vb.ranks:
        r.eng.profile(1)
    y, _, _ = vb.step(x)
    for r in vb.ranks:
        r.eng.profile_reset()
    vb.exchanged_bytes = 0
    for _ in range(K):
        y, _, _ = vb.step(x)
    torch.cuda.synchronize()
    per_rank, xch = [], []
    for r in vb.ranks:
        st = r.eng.profile_read()
        per_rank.append(sum(k["ms"] for k in st) / K)
        xch.append(sum(k["ms"] for k in st if k["name"].startswith("band_")) / K)
    if os.environ.get("BAND_CLASSES"):
        slow_r = max(range(n), key=lambda i: per_rank[i])
        rows = sorted(vb.ranks[slow_r].eng.profile_read(), key=lambda k: -k["ms"])
        print(f"     slowest rank {slow_r} by class: " + ", ".join(f"{k['name']} {k['ms'] / K:.3f} ({k['launches'] // K})" for k in rows[:14]))
    err = (y - y0).abs().max().item() / y0.abs().max().item()
    slow = max(per_rank)
    print(f"  n={n}: rows {vb.starts}")
    print(f"     per-rank kernel ms {[round(t, 3) for t in per_rank]} (of which pack/unpack {[round(t, 3) for t in xch]})")
    print(f"     slowest rank {slow:.3f} ms -> compute-only speed-up bound {t_ref / slow:.2f}x of {n} "
          f"(sum over ranks {sum(per_rank):.3f} ms = {sum(per_rank) / t_ref:.2f}x the unsharded work)")
    print(f"     {vb.ranks[0].num_exchanges} exchanges / step, {vb.exchanged_bytes / K / n / 1e6:.1f} MB sent per rank per step; "
          f"max|y - y_unsharded| / max|y| = {err:.2e}")
    # comm / compute overlap on ONE GPU: the same world with the exchanges' device copies on a side stream (what an RCCL transport
    # stream does on a node); wall time of whole steps, profiling off, sync vs async copies
    import time
    walls = {}
    for mode in ("sync", "split", "async"):
        if mode == "split":
            os.environ["WX_BAND_SPLIT"] = "1"
        w = vb if mode == "sync" else VirtualBands(cfg, sd, n, prec, async_copies=(mode == "async"))
        os.environ.pop("WX_BAND_SPLIT", None)
        if mode == "split":   # what the split itself costs every rank in kernel time
            for r in w.ranks:
                r.eng.profile(1)
            w.step(x)
            for r in w.ranks:
                r.eng.profile_reset()
            for _ in range(K):
                w.step(x)
            torch.cuda.synchronize()
            split_rank = [sum(k["ms"] for k in r.eng.profile_read()) / K for r in w.ranks]
            print(f"     interior / boundary split of the six decoder convolutions: per-rank kernel ms {[round(t, 3) for t in split_rank]} "
                  f"(slowest {max(split_rank):.3f} vs {slow:.3f} unsplit)")
        for r in w.ranks:
            r.eng.profile(0)
        w.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            w.step(x)
        torch.cuda.synchronize()
        walls[mode] = (time.perf_counter() - t0) / K * 1e3
        if mode != "sync":
            del w
    print(f"     whole-world wall per step (all {n} virtual ranks back to back on this GPU, profiling off): unsplit, copies on the compute "
          f"stream {walls['sync']:.3f} ms | split, compute stream {walls['split']:.3f} ms | split, copies on a side stream between events "
          f"{walls['async']:.3f} ms ({2 * n * vb.ranks[0].num_exchanges} cross-stream event edges per step)")
    out["bands"].append({"n": n, "rows": vb.starts, "wall_sync_ms": walls["sync"], "wall_split_ms": walls["split"], "wall_async_ms": walls["async"], "per_rank_ms": per_rank, "pack_unpack_ms": xch, "slowest_ms": slow,
                         "exchanges_per_step": vb.ranks[0].num_exchanges, "sent_MB_per_rank": vb.exchanged_bytes / K / n / 1e6,
                         "rel_err_vs_unsharded": err})
    del vb
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"band_time_{name}_{prec}.json"), "w"), indent=1)
