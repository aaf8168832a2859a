cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or parity or variants" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
B="python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in 1 2; do
$B 2>&1 | tail -1 | python -c "$P" steps4
WX_CONV3_SPLIT_STEPS=1000 $B 2>&1 | tail -1 | python -c "$P" off
WX_CONV3_SPLIT_STEPS=8 $B 2>&1 | tail -1 | python -c "$P" steps8
WX_CONV3_SPLIT_STEPS=2 $B 2>&1 | tail -1 | python -c "$P" steps2
done
python tools/stage_classes.py C1 bf16 2>&1 | grep "conv3\|gn_\|kernel time"
