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


def load(d, counter):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def family(name):
    name = name.replace("void ", "")
    for key in ("conv_gemm_dma_kernel", "conv_gemm_kernel", "ff_fused_kernel", "window_attn_kernel", "embed_patch_kernel"):
        if key in name:
            return "wx::" + key
    return name.split("(")[0][:60]


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    fam = defaultdict(lambda: {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
    for k, (n, v) in fetch.items():
        fam[family(k)]["launches"] += n
        fam[family(k)]["fetch_kb"] += v
    for k, (n, v) in write.items():
        fam[family(k)]["write_kb"] += v
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 "
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


if __name__ == "__main__":
    main()
