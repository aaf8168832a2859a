cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for b in attn_block_probe attn_block_probe_u2; do for a in "400 800 128 10 0 2" "200 400 256 10 0 2"; do timeout 120 tools/_build/$b $a | grep -v "dbg=[1247]"; done; done) > gpurun_out/j18_ab.log 2>&1
cat gpurun_out/j18_ab.log
