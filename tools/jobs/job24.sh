cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j24_kt -o kt -- python tools/fuxi_time.py bf16 5 > gpurun_out/j24_fuxi.log 2>&1
python tools/prof_summary.py gpurun_out/j24_kt > gpurun_out/j24_fuxi_kernel_stats.txt
rm -rf gpurun_out/j24_kt
tail -2 gpurun_out/j24_fuxi.log; head -30 gpurun_out/j24_fuxi_kernel_stats.txt
