cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/j50_pytest.log 2>&1
grep "passed\|failed" gpurun_out/j50_pytest.log
for i in 1 2 3; do for v in 2 0; do WX_ATTN_BLOCK=$v python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 WX_ATTN_BLOCK=$v', d['value'], d['ms_per_step'])"; done; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
