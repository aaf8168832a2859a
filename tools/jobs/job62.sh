cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in 1 2; do
$B 2>&1 | tail -1 | python -c "$P" new
WX_ALLOW_STALE=1 WX_LIBRARY=$GRAFT_REPO_ROOT/miles-credit_amd/wxengine/libwxengine_prev.so $B 2>&1 | tail -1 | python -c "$P" prev
done
WX_NO_EMBED_MERGE=1 $B 2>&1 | tail -1 | python -c "$P" nomerge
WX_GN_FOLD_TILES=0 $B 2>&1 | tail -1 | python -c "$P" nognfold
WX_GN_FOLD_TILES=256 $B 2>&1 | tail -1 | python -c "$P" gnfold256
python tools/stage_classes.py C1 bf16 2>&1 | grep -v amdgpu | head -40
