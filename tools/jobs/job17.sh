cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_variants_gpu.py tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -30) > gpurun_out/j17_pytest.log 2>&1
tail -30 gpurun_out/j17_pytest.log
(timeout 600 python bench.py --no-cpu-baseline --no-config2 2>&1 | tail -3) > gpurun_out/j17_bench.log 2>&1
(WX_NO_ATTN_BLOCK=1 timeout 600 python bench.py --no-cpu-baseline --no-config2 2>&1 | tail -3) > gpurun_out/j17_bench_off.log 2>&1
python - <<'PY'
import json
for f in ('gpurun_out/j17_bench.log','gpurun_out/j17_bench_off.log'):
  for l in open(f):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r['frac'], r['by_class_ms_per_step'])
PY
