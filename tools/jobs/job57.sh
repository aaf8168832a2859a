cd $GRAFT_REPO_ROOT
python tools/stage_classes.py C1 bf16 2>&1 | grep -v amdgpu | head -45
