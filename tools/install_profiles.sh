#!/bin/bash
# Build container: copy ONE collection call's outputs (gpurun_out/<tag>_*, tools/collect_r06.sh) under the names profiles/ keeps.
#   bash tools/install_profiles.sh r06
tag=${1:-r06}
g=gpurun_out
cp $g/${tag}_bench.json profiles/bench_${tag}.json
cp $g/${tag}_bench_config2_1deg.json profiles/bench_${tag}_config2_1deg.json
cp $g/${tag}_bench_events_off.json profiles/bench_${tag}_events_off.json
cp $g/${tag}_bench_fp32s.json profiles/bench_${tag}_fp32s.json
cp $g/${tag}_pmc_traffic.json profiles/pmc_traffic_${tag}.json
[ -f $g/${tag}_kernel_time.json ] && cp $g/${tag}_kernel_time.json profiles/kernel_time_${tag}.json
cp $g/${tag}_pmc_traffic.txt profiles/${tag}_pmc_traffic_by_shape.txt
cp $g/${tag}_kernel_stats.txt profiles/${tag}_rocprofv3_kernel_stats.txt
[ -f $g/${tag}_fp32s_kernel_stats.txt ] && cp $g/${tag}_fp32s_kernel_stats.txt profiles/${tag}_fp32s_rocprofv3_kernel_stats.txt
cp $g/${tag}_fuxi_kernel_stats.txt profiles/${tag}_fuxi_rocprofv3_kernel_stats.txt
cp $g/${tag}_utilisation.txt profiles/${tag}_utilisation_mfma_hbm.txt
for f in ab_schedules ab_gemm8p gemm8p_probe vendor_gemm_calibration fuxi_forward latband_virtual_ranks_C3_bf16 split_accuracy stage_classes_C3_bf16 stage_classes_C3_fp32 stage_classes_C3_fp32s; do
  [ -f $g/${tag}_$f.txt ] && cp $g/${tag}_$f.txt profiles/${tag}_$f.txt
done
[ -f $g/${tag}_gpu_suite.txt ] && cp $g/${tag}_gpu_suite.txt profiles/${tag}_gpu_suite.txt
git status --short profiles | head -30
