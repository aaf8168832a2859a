cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 2 1; do WX_ATTN_BLOCK=$v python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 WX_ATTN_BLOCK=$v', d['value'], d['ms_per_step'])"; done; done
WX_ATTN_BLOCK=1 python tools/stage_classes.py C1 bf16 2>&1 | grep "kernel time\|attn_block\|ff_fused\|gemm_ff1.s[12]\|gemm_ff2.s[12]" 
