cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/j14_pytest.log 2>&1
head -3 gpurun_out/j14_pytest.log
for i in 1 2; do
for lib in libwxengine_prev.so libwxengine.so; do
  WX_LIBRARY=$PWD/miles-credit_amd/wxengine/$lib timeout 300 python bench.py --no-cpu-baseline --no-fp32 --no-config2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']['by_class_ms_per_step']
print('$lib', d['value'], d['ms_per_step'], {k:r[k] for k in ('out_ff_qkv_fused','out_ff_fused','gemm_ff1','window_attn')})"
done; done
python - <<'PY'
import sys
sys.path[:0]=['miles-credit_amd','.']
import numpy as np, torch, os
from wxengine.config import named_config
from wxengine.engine import WXEngine
from wxengine.synth import synth_input, synth_state_dict
g=np.load('tests/golden/model_C3.npz'); s=int(g['stride'])
cfg=named_config('C3'); sd=synth_state_dict(cfg); x=torch.from_numpy(synth_input(cfg)).cuda()
e=WXEngine(cfg,'bf16',0); e.load_state_dict(sd); e.finalize()
y=e.forward(x)[0,:,0,::s,::s].cpu().numpy().astype(np.float64); r=g['y'].astype(np.float64)
print('C3 bf16 vs reference golden: rel-L2 %.4e max %.4e' % (np.linalg.norm(y-r)/np.linalg.norm(r), np.abs(y-r).max()/np.abs(r).max()))
PY
