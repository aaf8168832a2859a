cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -6
B="python bench.py --config C1 --steps 96 --warmup 12 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B > /dev/null 2>&1
for i in 1 2 3; do
WX_ALLOW_STALE=1 WX_LIBRARY=$GRAFT_REPO_ROOT/miles-credit_amd/wxengine/libwxengine_prev.so $B 2>&1 | tail -1 | python -c "$P" prev
$B 2>&1 | tail -1 | python -c "$P" new
done
python tools/stage_classes.py C1 bf16 2>&1 | grep "ff\|kernel time"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | cut -c1-160
