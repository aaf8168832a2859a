cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 4 6 12; do echo "WX_EMBED_SPLIT=$w"; WX_EMBED_SPLIT=$w timeout 300 python tools/stage_classes.py C1 bf16 embed_patch; WX_EMBED_SPLIT=$w timeout 300 python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-fp32 --no-config2 --no-roofline 2>&1 | tail -1 | cut -c1-160; done > gpurun_out/j29.txt 2>&1
cat gpurun_out/j29.txt
