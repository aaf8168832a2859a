cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j20_kt -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fp32 --no-config2 > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/j20_kt > gpurun_out/j20_kernel_stats.txt
rm -rf gpurun_out/j20_kt
head -30 gpurun_out/j20_kernel_stats.txt
