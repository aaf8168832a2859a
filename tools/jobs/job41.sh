cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_fx -o kt -- python tools/fuxi_time.py bf16 5 > gpurun_out/r03_fuxi_time.log 2>&1
python tools/prof_summary.py gpurun_out/r03_fx > gpurun_out/r03_fuxi_kernel_stats.txt
rm -rf gpurun_out/r03_fx
python tools/fuxi_time.py bf16 10 2>&1 | tail -1 > gpurun_out/r03_fuxi_forward.txt
python tools/fuxi_time.py fp32 3 2>&1 | tail -1 >> gpurun_out/r03_fuxi_forward.txt
cat gpurun_out/r03_fuxi_forward.txt; head -14 gpurun_out/r03_fuxi_kernel_stats.txt
