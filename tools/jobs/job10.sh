cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(WX_ABL=1 WX_ONLY=0 timeout 900 tools/_build/gemm_s32_probe 0 2>&1) > gpurun_out/j10_abl.log 2>&1
(WX_ONLY=1 timeout 900 tools/_build/gemm_s32_probe 0 2>&1) >> gpurun_out/j10_abl.log 2>&1
(WX_ONLY=4 timeout 900 tools/_build/gemm_s32_probe 0 2>&1) >> gpurun_out/j10_abl.log 2>&1
grep -E "ablation|us .* TF|FAIL|PROBE|PIN" gpurun_out/j10_abl.log
