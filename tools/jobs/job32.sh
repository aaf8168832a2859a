cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for ab in 0 1; do echo "WX_ABLK=$ab"; WX_ABLK=$ab WX_QUICK=1 timeout 600 tools/_build/gemm_s32_probe 1 | grep "^s[0-9]" | cut -c1-75; done) > gpurun_out/j32.txt 2>&1
cat gpurun_out/j32.txt
