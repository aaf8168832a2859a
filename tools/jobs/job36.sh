cd $GRAFT_REPO_ROOT
(timeout 600 tools/_build/gemm_stream_probe 1) > gpurun_out/j36.txt 2>&1
grep "^s[0-9]\|col tiles" gpurun_out/j36.txt | cut -c1-230
