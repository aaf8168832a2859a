cd $GRAFT_REPO_ROOT
B="python bench.py --config C1 --steps 96 --warmup 12 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B > /dev/null 2>&1
for i in 1 2 3 4; do
WX_NO_ATTN_PACK2=1 $B 2>&1 | tail -1 | python -c "$P" off
$B 2>&1 | tail -1 | python -c "$P" new
done
