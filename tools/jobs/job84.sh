cd $GRAFT_REPO_ROOT
for fm in 5 4 3 2; do
echo "FM $fm"; WX_N128_FM_SMALL=$fm python tools/stage_classes.py C3 bf16 2>&1 | grep "gemm_out.s3\|gemm_ff2.s3\|kernel time"
done
for fm in 4 3 2; do
WX_N128_FM_SMALL=$fm timeout 600 python -m pytest tests -m gpu -x -q -k "full_size_vs_reference_golden and C3 and bf16 and not C3S" 2>&1 | grep -E "passed|failed" | tail -1
done
