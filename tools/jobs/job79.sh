cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or variants" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | cut -c1-330
