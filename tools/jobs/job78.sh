cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-fp32 --no-config2 2>&1 | tail -1 > gpurun_out/r03_c1_run$i.json
python -c "import json; d=json.load(open('gpurun_out/r03_c1_run$i.json')); print(d['value'], d['ms_per_step'])"
done
