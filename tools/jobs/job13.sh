cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['launches_per_step'], r['all_kernels_ms_per_step']); print(r['by_class_ms_per_step'])"
python - <<'PY'
import sys,os
sys.path[:0]=['miles-credit_amd','.']
import torch
from wxengine.config import named_config
from wxengine.engine import WXEngine
from wxengine.synth import synth_input, synth_state_dict
cfg=named_config('C1'); e=WXEngine(cfg,'bf16',0); e.load_state_dict(synth_state_dict(cfg)); e.finalize()
x=torch.from_numpy(synth_input(cfg)).cuda()
e.forward(x); e.profile(2); e.profile_reset()
for _ in range(5): e.forward(x)
torch.cuda.synchronize()
rows=sorted(e.profile_read(), key=lambda r:-r['ms'])
print(sum(r['launches'] for r in rows)//5, 'launches per forward')
for r in rows[:40]: print(f"{r['name']:28s} {r['launches']//5:3d} launches {r['ms']/5*1e3:7.1f} us  ({r['ms']/max(r['launches'],1)*1e3:5.1f} us each)")
PY
