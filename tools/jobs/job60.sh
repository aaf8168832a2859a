cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or parity or gemm" 2>&1 | tail -3
B="python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in 1 2; do
$B 2>&1 | tail -1 | python -c "$P" new
WX_SKINNY_MAX=0 $B 2>&1 | tail -1 | python -c "$P" off
done
WX_SKINNY_MAX=16 $B 2>&1 | tail -1 | python -c "$P" max16
WX_SKINNY_MAX=4 $B 2>&1 | tail -1 | python -c "$P" max4
WX_SKINNY_MIN_NK=8 $B 2>&1 | tail -1 | python -c "$P" minnk8
WX_SKINNY_MIN_NK=4 $B 2>&1 | tail -1 | python -c "$P" minnk4
WX_SKINNY_STEPS=1 WX_SKINNY_MAX=16 $B 2>&1 | tail -1 | python -c "$P" steps1max16
WX_SKINNY_TILES=64 WX_SKINNY_MIN_NK=8 $B 2>&1 | tail -1 | python -c "$P" tiles64nk8
python tools/stage_classes.py C1 bf16 2>&1 | grep -v amdgpu | head -24
