cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, 'miles-credit_amd')
from wxengine.config import named_config
from wxengine.engine import WXEngine
from wxengine.synth import synth_input, synth_state_dict
cfg = named_config('C3'); sd = synth_state_dict(cfg); x = torch.from_numpy(synth_input(cfg)).cuda()
g = np.load('tests/golden/model_C3.npz'); s = int(g['stride']); want = g['y'].astype(np.float64)
for mode in ('2', '0', '1'):
    os.environ['WX_ATTN_BLOCK'] = mode
    e = WXEngine(cfg, 'bf16', 0); e.load_state_dict(sd); e.finalize()
    y = e.forward(x).cpu()[0, :, 0, ::s, ::s].numpy().astype(np.float64)
    print('WX_ATTN_BLOCK=%s rel-L2 %.3e max %.3e (of %.3e)' % (mode, np.linalg.norm(y - want) / np.linalg.norm(want), np.abs(y - want).max(), np.abs(want).max()))
    del e
PY
