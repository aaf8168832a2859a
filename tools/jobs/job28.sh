cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/stage_classes.py C1 bf16 > gpurun_out/j28_c1.txt 2>&1
cat gpurun_out/j28_c1.txt | head -60
