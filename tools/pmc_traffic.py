#!/usr/bin/env python
"""Condense two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic_*.json.

Per MI355X_MICROARCH.md §HBM: the counters are collected in separate passes (TCC slots), FETCH_SIZE/WRITE_SIZE are in
KB per dispatch, and on gfx950 FETCH_SIZE tallies 128-byte requests as 64 B -> doubled here; WRITE_SIZE is uncalibrated.
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [kernel substring ...]
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def family(name):
    name = name.replace("void ", "")
    for key in ("gemm_stream_kernel", "gemm8p_kernel", "conv_gemm_dma_kernel", "conv_gemm_kernel", "ff_fused_kernel", "window_attn_kernel", "embed_patch_kernel"):
        if key in name:
            return "wx::" + key
    return name.split("(")[0][:60]


def load(d, counter, by_grid=False):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            if by_grid:
                k = (family(k) + " " + k.split("<")[1].split(">")[0][:40] if "<" in k else family(k), r.get("Grid_Size", "?"))
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    fam = defaultdict(lambda: {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
    for k, (n, v) in fetch.items():
        fam[family(k)]["launches"] += n
        fam[family(k)]["fetch_kb"] += v
    for k, (n, v) in write.items():
        fam[family(k)]["write_kb"] += v
    # the library these counters belong to: bench.py prints `traffic: null` + a stale note when the library it loaded has another hash
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miles-credit_amd"))
    try:
        import build as wx_build
        wxsrc = wx_build.built_hash()
    except Exception:
        wxsrc = None
    steps = int(os.environ.get("WX_PMC_STEPS", "4"))   # forecast steps the profiled command ran (bench.py --steps 3 --warmup 1)
    tot_f = sum(v["fetch_kb"] for v in fam.values() if True) * 2 * 1024
    tot_w = sum(v["write_kb"] for v in fam.values()) * 1024
    wx_f = sum(v["fetch_kb"] for k, v in fam.items() if k.startswith("wx::")) * 2 * 1024
    wx_w = sum(v["write_kb"] for k, v in fam.items() if k.startswith("wx::")) * 1024
    out = {"wxsrc": wxsrc, "steps_profiled": steps,
           "engine_bytes_per_step": round((wx_f + wx_w) / steps), "engine_fetch_bytes_per_step": round(wx_f / steps),
           "engine_write_bytes_per_step": round(wx_w / steps), "all_kernels_bytes_per_step": round((tot_f + tot_w) / steps),
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 "
                     "--warmup 1 --no-cpu-baseline --no-roofline",
           "corrections": "FETCH_SIZE x2 on gfx950 (128-byte requests tallied as 64 B); WRITE_SIZE uncalibrated; KB -> bytes x1024",
           "kernels": {}}
    for f, v in sorted(fam.items(), key=lambda kv: -(kv[1]["fetch_kb"] + kv[1]["write_kb"])):
        n = max(v["launches"], 1)
        out["kernels"][f] = {"launches": v["launches"], "fetch_bytes_per_launch": round(2 * 1024 * v["fetch_kb"] / n),
                             "write_bytes_per_launch": round(1024 * v["write_kb"] / n)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for f, v in list(out["kernels"].items())[:8]:
        print(f, v)
    # per (kernel variant, grid size): which launches over-fetch
    fg, wg = load(sys.argv[1], "FETCH_SIZE", True), load(sys.argv[2], "WRITE_SIZE", True)
    print("\nper (variant, grid): launches, fetch MB/launch (x2 corrected), write MB/launch")
    for k, (n, v) in sorted(fg.items(), key=lambda kv: -kv[1][1])[:40]:
        w = wg.get(k, [1, 0.0])
        print(f"  {k[0]:70s} grid {k[1]:>9s}  n={n:4d}  fetch {2 * v / n / 1024:9.1f}  write {w[1] / max(w[0], 1) / 1024:9.1f}")


if __name__ == "__main__":
    main()
