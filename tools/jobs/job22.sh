cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_swin.py -m gpu -x -q -k "attend" 2>&1 | tail -30) > gpurun_out/j22_pytest.log 2>&1
tail -30 gpurun_out/j22_pytest.log
