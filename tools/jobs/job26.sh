cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_fuxi.py tests/test_swin.py -m gpu -x -q -s 2>&1 | tail -25) > gpurun_out/j26_pytest.log 2>&1
tail -25 gpurun_out/j26_pytest.log
