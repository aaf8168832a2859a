cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 0 1 2; do WX_FF_VARIANT=$v python bench.py --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WX_FF_VARIANT=$v', d['value'], d['ms_per_step'])"; done; done
