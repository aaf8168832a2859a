cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "T0F" 2>&1 | tail -5
