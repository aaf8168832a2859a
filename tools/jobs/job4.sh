cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 bash tools/pmc_probe.sh tools/_build/gemm_s32_probe 1 gemm_s 2>&1) > gpurun_out/j4_pmc_ff1.log 2>&1
cat gpurun_out/j4_pmc_ff1.log
