cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 tools/_build/gemm_s32_probe 1 2>&1) > gpurun_out/j3_s32.log 2>&1
(WX_ABLK=1 WX_OBLK=1 timeout 600 tools/_build/gemm_s32_probe 0 2>&1) > gpurun_out/j3_s32_blk.log 2>&1
cat gpurun_out/j3_s32.log; cat gpurun_out/j3_s32_blk.log
