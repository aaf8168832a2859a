cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
for lib in libwxengine_prev.so libwxengine.so; do
  WX_LIBRARY=$PWD/miles-credit_amd/wxengine/$lib timeout 300 python bench.py --no-cpu-baseline --no-fp32 --no-config2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']['by_class_ms_per_step']
print('$lib', d['value'], d['ms_per_step'], {k:r[k] for k in ('gemm_conv3','gemm_embed','gemm_convT4','gemm_convT2','gemm_ff1')})"
done; done
