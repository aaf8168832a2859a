#!/usr/bin/env python
"""Condense rocprofv3 output (kernel-trace CSV / counter CSV) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    for cut in ("(wx::", "(const", "("):
        i = name.find(cut)
        if i > 0:
            name = name[:i]
            break
    return name[:90]


def kernel_stats(path, by_grid=False):
    rows = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            n = short(r.get("Kernel_Name") or r.get("Name") or "?")
            if by_grid:   # one row per launch shape: the same kernel serves several stages
                n = f"{n[:78]} g={r.get('Grid_Size') or r.get('Grid_Size_X', '?')}"
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            rows[n][0] += 1
            rows[n][1] += dur
    tot = sum(v[1] for v in rows.values())
    print(f"# {path}\n# total kernel time {tot / 1e3:.3f} ms over {sum(v[0] for v in rows.values())} launches")
    print(f"{'kernel':92s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:92s} {c:7d} {t:12.1f} {t / c:10.2f} {100 * t / tot:6.2f}")


def counter_stats(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            n = short(r.get("Kernel_Name") or "?")
            c = r.get("Counter_Name")
            v = float(r.get("Counter_Value") or 0)
            acc[n][c][0] += 1
            acc[n][c][1] += v
    print(f"# {path}")
    for n, cs in sorted(acc.items()):
        for c, (k, v) in cs.items():
            print(f"{n:92s} {c:14s} dispatches={k:6d} sum={v:.6g} avg={v / k:.6g}")


def kernel_time_json(root, out_path, steps):
    """Per kernel family: launches and microseconds per forecast step, stamped with the library hash (bench.py reads it for
    roofline.frac_rocprof: HIP events around every launch inflate the in-process kernel time by 5-8 %)."""
    import json
    fam = defaultdict(lambda: [0, 0.0])
    for p in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        with open(p) as f:
            for r in csv.DictReader(f):
                n = (r.get("Kernel_Name") or "?").replace("void ", "")
                key = n.split("<")[0].split("(")[0]
                fam[key][0] += 1
                fam[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miles-credit_amd"))
    try:
        import build as wx_build
        wxsrc = wx_build.built_hash()
    except Exception:
        wxsrc = None
    gemm = ("wx::gemm_stream_kernel", "wx::conv_gemm_dma_kernel", "wx::gemm8p_kernel", "wx::gemm_wreg_kernel", "wx::conv_gemm_kernel")
    out = {"wxsrc": wxsrc, "steps_profiled": steps, "source": "rocprofv3 --kernel-trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline ...",
           "engine_kernel_us_per_step": round(sum(v[1] for k, v in fam.items() if k.startswith("wx::")) / steps, 1),
           "gemm_family_us_per_step": round(sum(v[1] for k, v in fam.items() if k in gemm) / steps, 1),
           "gemm_family_launches_per_step": round(sum(v[0] for k, v in fam.items() if k in gemm) / steps, 2),
           "window_attn_us_per_step": round(sum(v[1] for k, v in fam.items() if k == "wx::window_attn_kernel") / steps, 1),
           "families": {k: {"launches_per_step": round(v[0] / steps, 2), "us_per_step": round(v[1] / steps, 1)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]) if k.startswith("wx::")}}
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    root = sys.argv[1]
    if "--json" in sys.argv:
        i = sys.argv.index("--json")
        kernel_time_json(root, sys.argv[i + 1], int(sys.argv[i + 2]))
        sys.exit(0)
    for p in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        kernel_stats(p, by_grid="--by-grid" in sys.argv)
    for p in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        counter_stats(p)
