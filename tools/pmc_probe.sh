#!/bin/bash
# GPU box: SQ counter passes over one shape of a probe binary: tools/pmc_probe.sh <binary> <shape index> [kernel-name pattern]
bin=$1; idx=${2:-1}; pat=${3:-gemm_s}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  WX_ONLY=$idx WX_QUICK=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_probe_$i -o p -- $bin 0 > /dev/null 2>&1
done
python - "$pat" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float); nd = collections.Counter()
for d in sorted(glob.glob('gpurun_out/pmc_probe_*')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    for r in csv.DictReader(open(f[0])):
        if pat in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0].replace('void wx::', '')[:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
            if r.get('Start_Timestamp'):
                dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3; nd[k] += 1
names = sorted({c for k in agg for c in agg[k]})
for k in agg:
    print(f"{k}   avg {dur[k] / max(nd[k], 1):.1f} us (profiled)")
    wc = agg[k].get('SQ_WAVE_CYCLES', 0) / max(cnt[(k, 'SQ_WAVE_CYCLES')], 1)
    for c in names:
        if c in agg[k]:
            v = agg[k][c] / cnt[(k, c)]
            print(f"   {c:28s} {v:16.0f}" + (f"   {100 * v / wc:6.1f} % of wave cycles" if wc and c.startswith(('SQ_WAIT', 'SQ_ACTIVE', 'SQ_INST_CYCLES')) else ""))
PY
rm -rf gpurun_out/pmc_probe_*
