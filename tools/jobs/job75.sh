cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or full_size or rollout or embed or variants" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
B="python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in 1 2 3; do
$B 2>&1 | tail -1 | python -c "$P" new
WX_NO_EMBED_TAIL_SPLIT=1 $B 2>&1 | tail -1 | python -c "$P" off
done
python tools/stage_classes.py C3 bf16 2>&1 | grep "embed_patch\|kernel time"
WX_NO_EMBED_TAIL_SPLIT=1 python tools/stage_classes.py C3 bf16 2>&1 | grep "embed_patch\|kernel time"
