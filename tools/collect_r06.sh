#!/bin/bash
# GPU box, ONE call: everything profiles/ quotes for round 6 (-> gpurun_out/r06_*).   bash tools/collect_r06.sh
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
bash tools/util_report.sh r06 >> gpurun_out/r06_collect.log 2>&1
python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-fp32 --no-config2 --no-host-delivery --no-concurrent 2>&1 | tail -1 > gpurun_out/r06_bench_config2_1deg.json
python tools/stage_classes.py C3 bf16 > gpurun_out/r06_stage_classes_C3_bf16.txt 2>&1
python bench.py --precision fp32s --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-host-delivery --no-concurrent 2>&1 | tail -1 > gpurun_out/r06_bench_fp32s.json
# the eight-phase kernel: same-box A/B of its dispatch, the probe with ablations and phase stamps, the vendor GEMM on the same lease
python tools/ab_time.py --config C3 --arm gemm8p: --arm off:WX_NO_GEMM8P=1 --rounds 3 --steps 20 2>&1 | grep "\[ab\]" > gpurun_out/r06_ab_gemm8p.txt
timeout 600 tools/_build/gemm8p_probe 2 > gpurun_out/r06_gemm8p_probe.txt 2>&1
python tools/vendor_gemm_calib.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_vendor_gemm_calibration.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_fx -o kt -- python tools/fuxi_time.py bf16 5 > gpurun_out/r06_fuxi_time.log 2>&1
python tools/prof_summary.py gpurun_out/r06_fx > gpurun_out/r06_fuxi_kernel_stats.txt
rm -rf gpurun_out/r06_fx
python tools/fuxi_time.py bf16 10 2>&1 | tail -1 > gpurun_out/r06_fuxi_forward.txt
BAND_CLASSES=1 python tools/band_time.py C3 bf16 8 > gpurun_out/r06_latband_virtual_ranks_C3_bf16.txt 2>&1
cut -c1-500 gpurun_out/r06_bench.json; head -16 gpurun_out/r06_kernel_stats.txt; cat gpurun_out/r06_ab_gemm8p.txt
