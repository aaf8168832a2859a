cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_variants_gpu.py tests/test_rollout_gpu.py -m gpu -x -q -s 2>&1 | tail -60) > gpurun_out/j2_pytest.log 2>&1
(timeout 300 tools/_build/mfma_probe 2>&1) > gpurun_out/j2_mfma.log 2>&1
tail -15 gpurun_out/j2_pytest.log; tail -46 gpurun_out/j2_mfma.log
