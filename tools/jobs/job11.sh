cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_latband_gpu.py -m gpu -x -q 2>&1 | tail -15) > gpurun_out/j11_pytest.log 2>&1
(timeout 900 python tools/band_time.py C3 bf16 4 8 2>&1 | grep -v amdgpu.ids) > gpurun_out/j11_band_time.log 2>&1
tail -6 gpurun_out/j11_pytest.log; cat gpurun_out/j11_band_time.log
