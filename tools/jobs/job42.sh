cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/j42_pytest.log 2>&1
grep "passed\|failed" gpurun_out/j42_pytest.log
for i in 1 2 3; do
for v in 2 0; do WX_ATTN_BLOCK=$v python bench.py --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WX_ATTN_BLOCK=$v', d['value'], d['ms_per_step'])"; done; done
python bench.py --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['frac'], r['attention'], r.get('attention_block'))"
