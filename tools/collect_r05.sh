#!/bin/bash
# GPU box, ONE call: everything profiles/ quotes for round 5 (-> gpurun_out/r05_*).   bash tools/collect_r05.sh
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
bash tools/util_report.sh r05 >> gpurun_out/r05_collect.log 2>&1
python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-fp32 --no-config2 --no-host-delivery --no-concurrent 2>&1 | tail -1 > gpurun_out/r05_bench_config2_1deg.json
python tools/stage_classes.py C3 bf16 > gpurun_out/r05_stage_classes_C3_bf16.txt 2>&1
# the split-bf16 mode (fp32 storage): bench line of its own, per-class table beside the exact-f32 engine's, rocprofv3 kernel stats
python bench.py --precision fp32s --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-host-delivery --no-concurrent 2>&1 | tail -1 > gpurun_out/r05_bench_fp32s.json
python tools/stage_classes.py C3 fp32s > gpurun_out/r05_stage_classes_C3_fp32s.txt 2>&1
python tools/stage_classes.py C3 fp32 > gpurun_out/r05_stage_classes_C3_fp32.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_ks -o kt -- python bench.py --precision fp32s --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-config2 --no-host-delivery --no-concurrent > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/r05_ks > gpurun_out/r05_fp32s_kernel_stats.txt
rm -rf gpurun_out/r05_ks
python tools/split_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_split_accuracy.txt
# same-box A/B of the round's schedule switches
python tools/ab_time.py --config C3 --arm one: --arm two:WX_TWO_STREAM=1 --arm graph:WX_GRAPH=1 --rounds 3 --steps 20 --check-equal 2>&1 | grep "\[ab\]" > gpurun_out/r05_ab_schedules.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_fx -o kt -- python tools/fuxi_time.py bf16 5 > gpurun_out/r05_fuxi_time.log 2>&1
python tools/prof_summary.py gpurun_out/r05_fx > gpurun_out/r05_fuxi_kernel_stats.txt
rm -rf gpurun_out/r05_fx
python tools/fuxi_time.py bf16 10 2>&1 | tail -1 > gpurun_out/r05_fuxi_forward.txt
BAND_CLASSES=1 python tools/band_time.py C3 bf16 8 > gpurun_out/r05_latband_virtual_ranks_C3_bf16.txt 2>&1
cut -c1-400 gpurun_out/r05_bench.json; head -14 gpurun_out/r05_kernel_stats.txt; cut -c1-300 gpurun_out/r05_bench_fp32s.json; cat gpurun_out/r05_ab_schedules.txt
