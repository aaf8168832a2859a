cd $GRAFT_REPO_ROOT
for c in 0 1; do echo "WX_GEMM_CFG=$c"; WX_GEMM_CFG=$c python tools/stage_classes.py C3 bf16 gemm_ 2>&1 | grep -v amdgpu; done > gpurun_out/j35.txt 2>&1
python - <<'PY'
import re
blocks=open('gpurun_out/j35.txt').read().split('WX_GEMM_CFG=')[1:]
d=[]
for b in blocks:
    m={}
    for l in b.splitlines():
        t=l.split()
        if len(t)>=7 and t[0].startswith('gemm_'): m[t[0]]=(float(t[1]), float(t[6]))
    d.append(m)
for k in sorted(d[0], key=lambda k:-d[0][k][0]):
    print(f"{k:<18s} cfg0 {d[0][k][0]:.3f} ms ({d[0][k][1]:.1f} us)  cfg1 {d[1].get(k,(0,0))[0]:.3f} ms ({d[1].get(k,(0,0))[1]:.1f} us)")
PY
