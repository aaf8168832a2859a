#!/bin/bash
# GPU box, ONE call: everything profiles/ quotes for round 4 (-> gpurun_out/r04_*).   bash tools/collect_r04.sh
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
bash tools/util_report.sh r04 >> gpurun_out/r04_collect.log 2>&1
python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-fp32 --no-config2 --no-host-delivery --no-concurrent 2>&1 | tail -1 > gpurun_out/r04_bench_config2_1deg.json
python tools/stage_classes.py C3 bf16 > gpurun_out/r04_stage_classes_C3_bf16.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_fx -o kt -- python tools/fuxi_time.py bf16 5 > gpurun_out/r04_fuxi_time.log 2>&1
python tools/prof_summary.py gpurun_out/r04_fx > gpurun_out/r04_fuxi_kernel_stats.txt
rm -rf gpurun_out/r04_fx
python tools/fuxi_time.py bf16 10 2>&1 | tail -1 > gpurun_out/r04_fuxi_forward.txt
BAND_CLASSES=1 python tools/band_time.py C3 bf16 8 > gpurun_out/r04_latband_virtual_ranks_C3_bf16.txt 2>&1
tools/_build/gemm_wreg_probe 1 > gpurun_out/r04_gemm_wreg_probe_raw.txt 2>&1
WX_LC_CFG=2 tools/_build/gemm_lc_probe 0 > gpurun_out/r04_gemm_lc_probe_raw.txt 2>&1
python tools/stress_report.py > gpurun_out/r04_stress_report.txt 2>&1
cut -c1-400 gpurun_out/r04_bench.json; head -14 gpurun_out/r04_kernel_stats.txt; cat gpurun_out/r04_fuxi_forward.txt; cut -c1-200 gpurun_out/r04_bench_config2_1deg.json
