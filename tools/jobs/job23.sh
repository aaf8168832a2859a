cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(WX_ONLY=0 WX_ABL=1 timeout 600 tools/_build/gemm_pp_probe 0) > gpurun_out/j23_pp.log 2>&1
grep -v parity gpurun_out/j23_pp.log
