cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/j9_pytest.log 2>&1
(timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/j9_bench.log 2>&1
tail -5 gpurun_out/j9_pytest.log; python - <<'PY'
import json
for l in open('gpurun_out/j9_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['by_class_ms_per_step']); print(d['config2']['value'], d['config5'])
PY
