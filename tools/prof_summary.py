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


if __name__ == "__main__":
    root = sys.argv[1]
    for p in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        kernel_stats(p, by_grid="--by-grid" in sys.argv)
    for p in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        counter_stats(p)
