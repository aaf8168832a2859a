#!/bin/bash
# GPU box: SQ counter passes for kernels whose name contains $1 (python tools/gpu_time.py as the workload)
pat=${1:-window_attn}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_sq_$i -o p -- python tools/gpu_time.py --steps 1 > /dev/null 2>&1
done
python - "$pat" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
for d in sorted(glob.glob('gpurun_out/pmc_sq_*')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if pat in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0][:70] + ' grid ' + r.get('Grid_Size', '?')
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k in agg:
        print(k)
        for c, v in agg[k].items(): print(f"   {c:32s} {v / cnt[(k, c)]:16.0f}  (per dispatch, n={cnt[(k, c)]})")
PY
rm -rf gpurun_out/pmc_sq_*
