cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/j54_pytest.log 2>&1
grep "passed\|failed" gpurun_out/j54_pytest.log
python bench.py --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | cut -c60-150
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
