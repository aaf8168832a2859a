cd $GRAFT_REPO_ROOT
B="python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B 2>&1 | tail -1 | python -c "$P" default
HIP_FORCE_DEV_KERNARG=1 $B 2>&1 | tail -1 | python -c "$P" devkernarg1
HIP_FORCE_DEV_KERNARG=0 $B 2>&1 | tail -1 | python -c "$P" devkernarg0
GPU_MAX_HW_QUEUES=1 $B 2>&1 | tail -1 | python -c "$P" hwq1
HSA_ENABLE_INTERRUPT=0 $B 2>&1 | tail -1 | python -c "$P" nointerrupt
$B 2>&1 | tail -1 | python -c "$P" default
env | grep -i "HIP_\|HSA_\|GPU_\|ROC" | head
