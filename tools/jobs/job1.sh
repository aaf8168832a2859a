set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80) > gpurun_out/j1_pytest.log 2>&1
(timeout 600 python bench.py 2>&1 | tail -5) > gpurun_out/j1_bench.log 2>&1
(timeout 300 tools/_build/mfma_probe 2>&1) > gpurun_out/j1_mfma.log 2>&1
tail -5 gpurun_out/j1_pytest.log; cat gpurun_out/j1_mfma.log
