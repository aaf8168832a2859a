cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -5
bash tools/jobs/job82.sh
