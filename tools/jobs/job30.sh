cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(WX_QUICK=1 timeout 600 tools/_build/gemm_s32_probe 1) > gpurun_out/j30.txt 2>&1
grep "^s[0-9]\|^tail\|^tiny\|PROBE" gpurun_out/j30.txt | cut -c1-150
