cd $GRAFT_REPO_ROOT
B="python bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])'
$B 2>&1 | tail -1 | python -c "$P" new
WX_FF_MIN_WGS=40 $B 2>&1 | tail -1 | python -c "$P" ffmin40
WX_FF_MIN_WGS=20 $B 2>&1 | tail -1 | python -c "$P" ffmin20
WX_FF_MIN_WGS=1 $B 2>&1 | tail -1 | python -c "$P" ffmin1
$B 2>&1 | tail -1 | python -c "$P" new
WX_FF_MIN_WGS=40 python tools/stage_classes.py C1 bf16 2>&1 | grep -v amdgpu | grep "ff\|kernel time"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c1prof -o c1 -- python $GRAFT_REPO_ROOT/bench.py --config C1 --steps 48 --warmup 6 --no-cpu-baseline --no-config2 --no-fp32 --no-roofline > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/c1prof | head
