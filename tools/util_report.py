"""Per-kernel utilisation table from three rocprofv3 passes (see tools/util_report.sh).
  MFMA busy %  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 256 CUs x 4 SIMDs): matrix-pipe cycles over the pipe
                 cycles the launch had AT THE MAXIMUM CLOCK.  The chip sustains ~2.0 GHz under MFMA load (s_memtime),
                 so the true occupancy of a matrix-heavy kernel is up to 1.2x the printed figure; no per-kernel clock is
                 printed because none of the counters gives one (round 1 normalised by GRBM_GUI_ACTIVE, which also counts the
                 dispatch gap and read 2.6-9.2 "GHz" on short kernels).  The counter equals 16 x SQ_INSTS_MFMA for
                 v_mfma_f32_16x16x32_bf16.
  MFMA TF/s    = SQ_INSTS_MFMA * 16384 FLOP / duration  (what the matrix pipes executed, padded tiles included), vs 2500
  HBM GB/s     = (FETCH_SIZE KB * 2 [gfx950 tallies 128-B requests as 64 B] + WRITE_SIZE KB) * 1024 / duration, vs 8000"""
import collections
import csv
import glob
import sys


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(f[0])):
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return rows, dur


def short(name):
    n = name.split("(")[0]
    n = n.replace("unsigned short", "bf16")
    return n[:64]


def main():
    a, da = load(sys.argv[1])
    f, df = load(sys.argv[2])
    w, dw = load(sys.argv[3])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for (did, k), c in a.items():
        g = agg[short(k)]
        g["n"] += 1
        g["dur"] += da.get((did, k), 0.0)
        for cn, v in c.items():
            g[cn] += v
    for src, dd, cname in ((f, df, "FETCH_SIZE"), (w, dw, "WRITE_SIZE")):
        for (did, k), c in src.items():
            g = agg[short(k)]
            g[cname] += c.get(cname, 0.0)
            g[cname + "_dur"] += dd.get((did, k), 0.0)
    tot = sum(g["dur"] for g in agg.values())
    print(f"# C3 bf16 benchmark steps under rocprofv3 --pmc (profiled clocks run ~3-5 % below un-profiled ones); total kernel time {tot * 1e3:.1f} ms")
    print(f"{'kernel':66s} {'calls':>6s} {'avg us':>8s} {'time %':>7s} {'MFMA busy % (>=)':>16s} {'MFMA TF/s':>10s} {'of 2500':>8s} {'HBM GB/s':>9s} {'of 8000':>8s}")
    for k, g in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
        if g["dur"] <= 0 or g["n"] < 1:
            continue
        busy = 100.0 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (g["dur"] * 2.4e9 * 1024)
        tfs = g["SQ_INSTS_MFMA"] * 16384 / g["dur"] / 1e12
        fb = g["FETCH_SIZE"] * 1024 * 2 / g["FETCH_SIZE_dur"] if g["FETCH_SIZE_dur"] else 0
        wb = g["WRITE_SIZE"] * 1024 / g["WRITE_SIZE_dur"] if g["WRITE_SIZE_dur"] else 0
        gbs = (fb + wb) / 1e9
        print(f"{k:66s} {int(g['n']):6d} {g['dur'] / g['n'] * 1e6:8.1f} {100 * g['dur'] / tot:7.2f} {busy:16.1f} {tfs:10.0f} {100 * tfs / 2500:7.1f}% {gbs:9.0f} {100 * gbs / 8000:7.1f}%")


if __name__ == "__main__":
    main()
