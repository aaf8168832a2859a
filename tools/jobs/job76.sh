cd $GRAFT_REPO_ROOT
python tools/stage_classes.py C3 bf16 2>&1 | grep -v amdgpu > gpurun_out/sc_def.txt
WX_GEMM_CFG=3 python tools/stage_classes.py C3 bf16 2>&1 | grep -v amdgpu > gpurun_out/sc_cfg3.txt
python - <<'P'
def rd(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)>=6 and p[2]=='ms': d[p[0]]=(float(p[1]),int(p[3]),float(p[5]))
    return d
a=rd('gpurun_out/sc_def.txt'); b=rd('gpurun_out/sc_cfg3.txt')
for k in sorted(a, key=lambda k:-a[k][0]):
    if k in b and abs(a[k][2]-b[k][2])>0.03*a[k][2]: print(f"{k:24s} def {a[k][2]:8.1f} us  cfg3 {b[k][2]:8.1f} us  x{a[k][1]}")
P
