cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 256 0 16 32; do WX_FF_MIN_WGS=$v python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 WX_FF_MIN_WGS=$v', d['value'], d['ms_per_step'])"; done; done
