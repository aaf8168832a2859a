cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_fuxi.py -m gpu -x -q -s 2>&1 | tail -6) > gpurun_out/j40.log 2>&1; tail -6 gpurun_out/j40.log
