cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/j27_pytest.log 2>&1
tail -6 gpurun_out/j27_pytest.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/j27_bench.json 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/j27_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r.get('traffic'))
print(r['by_class_ms_per_step'])
print('fp32', d['fp32']); print('cfg2', d['config2']); print('cfg5', d['config5']); print('cpu', d['cpu_baseline'])
PY
