cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for a in "400 800 128 10 0 2" "400 800 128 10 1 2"; do timeout 120 tools/_build/attn_block_probe $a; done) > gpurun_out/j18_ab.log 2>&1
cat gpurun_out/j18_ab.log
