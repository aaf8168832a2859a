#!/bin/bash
# GPU box: bench line + rocprofv3 kernel stats + HBM-traffic PMC passes for the current build -> gpurun_out/<tag>_*
tag=${1:-r01}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# counters FIRST, and into profiles/ of this copy: the bench line below then carries roofline.traffic of THIS library (bench.py checks the hash)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pf -o pf -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32 --no-config2 --no-host-delivery --no-concurrent > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pw -o pw -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32 --no-config2 --no-host-delivery --no-concurrent > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/${tag}_pf gpurun_out/${tag}_pw gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt; rm -rf gpurun_out/${tag}_pf gpurun_out/${tag}_pw
cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic_${tag}.json
# kernel trace BEFORE the bench line too: the line then carries roofline.frac_rocprof of this library (7 forecast steps = 5 + 2 warm-up)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32 --no-config2 --no-host-delivery --no-concurrent > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/${tag}_kt > gpurun_out/${tag}_kernel_stats.txt
python tools/prof_summary.py gpurun_out/${tag}_kt --json gpurun_out/${tag}_kernel_time.json 7
cp gpurun_out/${tag}_kernel_time.json profiles/kernel_time_${tag}.json
rm -rf gpurun_out/${tag}_kt
python bench.py --steps 40 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench.json
# same-box pair for profiles/README.md: un-profiled wall (events off) vs the rocprofv3 kernel sum above
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-fp32 --no-config2 --no-host-delivery --no-concurrent 2>&1 | tail -1 > gpurun_out/${tag}_bench_events_off.json
cut -c1-300 gpurun_out/${tag}_bench.json; head -12 gpurun_out/${tag}_kernel_stats.txt
