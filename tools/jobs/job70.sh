cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_c1 -o kt -- python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline > gpurun_out/r03_c1_bench_under_rocprof.log 2>&1
python tools/prof_summary.py gpurun_out/r03_c1 > gpurun_out/r03_c1_kernel_stats.txt
python - <<'P' >> gpurun_out/r03_c1_kernel_stats.txt
import csv, glob, statistics
f = glob.glob('gpurun_out/r03_c1/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
tail = rows[-2400:]
gaps = [tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)]
durs = [e - s for s, e, _ in tail]
print(f"# last {len(tail)} launches: median gap between consecutive kernels {statistics.median(gaps)} ns, shortest kernel {min(durs)} ns, median kernel {statistics.median(durs)} ns")
P
rm -rf gpurun_out/r03_c1
tail -3 gpurun_out/r03_c1_kernel_stats.txt
