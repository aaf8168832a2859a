cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 tools/_build/mfma_probe 2>&1) > gpurun_out/j5_mfma.log 2>&1
grep -A80 "MFMA waves exit random" gpurun_out/j5_mfma.log | grep -E "s_add|ds_read|s_nop|0 v_fma"
