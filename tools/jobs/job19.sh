cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python tools/stage_classes.py C3 bf16; WX_NO_ATTN_BLOCK=1 timeout 300 python tools/stage_classes.py C3 bf16) > gpurun_out/j19.log 2>&1
grep -v "^$" gpurun_out/j19.log | head -90
