cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in 0 1; do echo "WX_FF_VARIANT=$v"; WX_FF_VARIANT=$v python bench.py --no-cpu-baseline --no-config2 --no-fp32 --no-roofline 2>&1 | tail -1 | cut -c60-140; done; done
for v in 0 1; do WX_FF_VARIANT=$v python tools/stage_classes.py C3 bf16 ff_ 2>&1 | grep "ff_"; done
