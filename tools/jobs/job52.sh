cd $GRAFT_REPO_ROOT
python tools/stage_classes.py C1 bf16 2>&1 | grep "kernel time\|attn_block\|window_attn\|gemm_qkv\|gemm_out"
