cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or block or parity or variants or merge" 2>&1 | tail -5
for i in 1 2 3; do
python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])"
WX_ALLOW_STALE=1 WX_LIBRARY=$GRAFT_REPO_ROOT/miles-credit_amd/wxengine/libwxengine_prev.so python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['ms_per_step'])"
done
python tools/stage_classes.py C1 bf16 2>&1 | grep -v amdgpu | head -45
