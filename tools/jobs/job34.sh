cd $GRAFT_REPO_ROOT
for c in 0 1 2; do echo cfg=$c; WX_FUXI_CONV_CFG=$c python tools/fuxi_time.py bf16 10 2>&1 | tail -1; done
