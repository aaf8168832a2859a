cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_variants_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/j31_pytest.log 2>&1
tail -3 gpurun_out/j31_pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | cut -c1-140
WX_ALLOW_STALE=1 WX_LIBRARY=$GRAFT_REPO_ROOT/miles-credit_amd/wxengine/libwxengine_prev.so python bench.py --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | cut -c1-140
done
python tools/fuxi_time.py bf16 10 2>&1 | tail -1
