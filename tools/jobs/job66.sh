cd $GRAFT_REPO_ROOT
B="python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B 2>&1 | tail -1 | python -c "$P" split4
WX_EMBED_SPLIT=5 $B 2>&1 | tail -1 | python -c "$P" split5
WX_EMBED_SPLIT=10 $B 2>&1 | tail -1 | python -c "$P" split10
WX_EMBED_SPLIT=3 $B 2>&1 | tail -1 | python -c "$P" split3
WX_EMBED_SPLIT=2 $B 2>&1 | tail -1 | python -c "$P" split2
$B 2>&1 | tail -1 | python -c "$P" split4
WX_EMBED_SPLIT=5 python tools/stage_classes.py C1 bf16 2>&1 | grep "embed_patch"
WX_EMBED_SPLIT=10 python tools/stage_classes.py C1 bf16 2>&1 | grep "embed_patch"
