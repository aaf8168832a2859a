cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "profile or variants or fused or latband or band" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], r["frac"], r["avg_launch_us"], r["kernel_ms_per_step"], r["all_kernels_ms_per_step"], r["attention"]["frac"])'
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "$P" chained
WX_PROFILE_EVENT_PAIRS=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "$P" pairs
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "$P" chained
