cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or block or variants or parity" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -6
B="python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
for i in 1 2 3; do
$B 2>&1 | tail -1 | python -c "$P" new
WX_NO_ATTN_PACK2=1 $B 2>&1 | tail -1 | python -c "$P" off
done
python tools/stage_classes.py C1 bf16 2>&1 | grep "attn\|qkv.s2\|out.s2\|kernel time"
