cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 tools/_build/gemm_s32_probe 0 2>&1) > gpurun_out/j6_s32.log 2>&1
grep -E "us .* TF|FAIL|PROBE" gpurun_out/j6_s32.log
