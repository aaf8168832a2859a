cd $GRAFT_REPO_ROOT
for i in 1 2; do
for g in 0 1; do WX_GRAPH=$g python bench.py --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WX_GRAPH=$g', d['value'], d['ms_per_step'])"; done; done
for g in 0 1; do WX_GRAPH=$g python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 WX_GRAPH=$g', d['value'], d['ms_per_step'])"; done
