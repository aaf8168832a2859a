cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_variants_gpu.py -m gpu -x -q -k "attention_block" 2>&1 | tail -30) > gpurun_out/j21_pytest.log 2>&1
tail -30 gpurun_out/j21_pytest.log
