#!/bin/bash
# GPU box: MFMA utilisation, effective clock and HBM GB/s per kernel of the benchmark step (C3 bf16), against gfx950 peaks.
#   pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA   pass B/C: FETCH_SIZE / WRITE_SIZE
# (separate --pmc passes, --kernel-trace only: MI355X_MICROARCH.md "HBM / rocprofv3")
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
W="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32 --no-config2 --no-host-delivery --no-concurrent"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d gpurun_out/${tag}_ua -o ua -- $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_uf -o uf -- $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_uw -o uw -- $W > /dev/null 2>&1
python tools/util_report.py gpurun_out/${tag}_ua gpurun_out/${tag}_uf gpurun_out/${tag}_uw > gpurun_out/${tag}_utilisation.txt
rm -rf gpurun_out/${tag}_ua gpurun_out/${tag}_uf gpurun_out/${tag}_uw
cat gpurun_out/${tag}_utilisation.txt
