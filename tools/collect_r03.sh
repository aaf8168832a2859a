#!/bin/bash
# GPU box, ONE call: everything profiles/ quotes for round 3 (-> gpurun_out/r03_*).   bash tools/collect_r03.sh
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1
bash tools/util_report.sh r03 >> gpurun_out/r03_collect.log 2>&1
python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-fp32 --no-config2 2>&1 | tail -1 > gpurun_out/r03_bench_config2_1deg.json
python tools/stage_classes.py C3 bf16 > gpurun_out/r03_stage_classes_C3_bf16.txt 2>&1
WX_ATTN_BLOCK=1 python tools/stage_classes.py C3 bf16 > gpurun_out/r03_stage_classes_C3_bf16_attn_block.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_fx -o kt -- python tools/fuxi_time.py bf16 5 > gpurun_out/r03_fuxi_time.log 2>&1
python tools/prof_summary.py gpurun_out/r03_fx > gpurun_out/r03_fuxi_kernel_stats.txt
rm -rf gpurun_out/r03_fx
python tools/fuxi_time.py bf16 10 2>&1 | tail -1 > gpurun_out/r03_fuxi_forward.txt
python tools/fuxi_time.py fp32 3 2>&1 | tail -1 >> gpurun_out/r03_fuxi_forward.txt
(for a in "400 800 128 10 0 2" "400 800 128 10 1 2" "200 400 256 10 0 2" "200 400 256 5 1 2"; do tools/_build/attn_block_probe $a; done) > gpurun_out/r03_attn_block_probe.txt 2>&1
(WX_ABL=1 tools/_build/gemm_pp_probe 0) > gpurun_out/r03_gemm_pp_probe.txt 2>&1
cut -c1-400 gpurun_out/r03_bench.json; head -14 gpurun_out/r03_kernel_stats.txt; cat gpurun_out/r03_fuxi_forward.txt; cut -c1-200 gpurun_out/r03_bench_config2_1deg.json
