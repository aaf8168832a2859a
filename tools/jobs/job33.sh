cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export P=$GRAFT_REPO_ROOT/miles-credit_amd/wxengine/libwxengine_pair.so
(WX_LIBRARY=$P timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "full_size or every_block" 2>&1 | tail -4) > gpurun_out/j33_pytest.log 2>&1
tail -3 gpurun_out/j33_pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['value'], d['roofline']['attention']['frac'], d['roofline']['by_class_ms_per_step']['window_attn'])"
WX_LIBRARY=$P python bench.py --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair', d['value'], d['roofline']['attention']['frac'], d['roofline']['by_class_ms_per_step']['window_attn'])"
done
