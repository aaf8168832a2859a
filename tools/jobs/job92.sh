cd $GRAFT_REPO_ROOT
B="python bench.py --config C1 --steps 96 --warmup 12 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B > /dev/null 2>&1
for i in 1 2 3; do
$B 2>&1 | tail -1 | python -c "$P" px128
WX_FF_PX64=1 $B 2>&1 | tail -1 | python -c "$P" px64
done
WX_FF_PX64=1 python tools/stage_classes.py C1 bf16 2>&1 | grep "ff_fused.s1"
WX_FF_PX64=1 timeout 600 python -m pytest tests -m gpu -x -q -k "golden and C1" 2>&1 | grep -E "passed|failed" | tail -1
