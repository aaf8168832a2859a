cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
python tools/stage_classes.py C1 bf16 2>&1 | grep -v amdgpu | head -12
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | cut -c1-200
WX_ALLOW_STALE=1 WX_LIBRARY=$GRAFT_REPO_ROOT/miles-credit_amd/wxengine/libwxengine_prev.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | cut -c1-200
