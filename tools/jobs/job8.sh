cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_swin.py tests/test_variants_gpu.py -m gpu -x -q -s 2>&1 | tail -40) > gpurun_out/j8_pytest.log 2>&1
cat gpurun_out/j8_pytest.log
