cd $GRAFT_REPO_ROOT
for c in 0 1024 3072; do echo n128_max=$c; WX_SWIN_N128_MAX=$c python tools/fuxi_time.py bf16 10 2>&1 | tail -1; done
