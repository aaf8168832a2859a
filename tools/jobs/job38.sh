cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/j38_pytest.log 2>&1
tail -2 gpurun_out/j38_pytest.log | head -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
