cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_variants_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8
