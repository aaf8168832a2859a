cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(WX_FF_VARIANT=3 WX_NO_FFOUT=1 timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "fused_feed_forward or full_size or every_block" 2>&1 | tail -4) > gpurun_out/j48_pytest.log 2>&1
tail -3 gpurun_out/j48_pytest.log
for i in 1 2 3; do for v in 0 3; do WX_FF_VARIANT=$v python bench.py --no-cpu-baseline --no-config2 --no-fp32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']['by_class_ms_per_step']; print('WX_FF_VARIANT=$v', d['value'], d['ms_per_step'], r.get('ff_fused'))"; done; done
