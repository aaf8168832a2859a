cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_variants_gpu.py tests/test_engine_gpu.py tests/test_swin.py tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/j44_pytest.log 2>&1
grep "passed\|failed" gpurun_out/j44_pytest.log
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('new ', d['value'], d['ms_per_step'], r['attention']['frac'], r['by_class_ms_per_step']['window_attn'], r['by_class_ms_per_step']['attn_block'])"
WX_ALLOW_STALE=1 WX_LIBRARY=$GRAFT_REPO_ROOT/miles-credit_amd/wxengine/libwxengine_prev.so python bench.py --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('prev', d['value'], d['ms_per_step'], r['attention']['frac'], r['by_class_ms_per_step']['window_attn'], r['by_class_ms_per_step']['attn_block'])"
done
