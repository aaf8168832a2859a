cd $GRAFT_REPO_ROOT
B="python bench.py --config C1 --steps 96 --warmup 12 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B > /dev/null 2>&1
for i in 1 2; do
$B 2>&1 | tail -1 | python -c "$P" default
WX_FF64_VARIANT=3 $B 2>&1 | tail -1 | python -c "$P" c64px64
WX_FF64_VARIANT=4 $B 2>&1 | tail -1 | python -c "$P" c64px256
done
