cd $GRAFT_REPO_ROOT
bash tools/jobs/job78.sh
python tools/stage_classes.py C1 bf16 > gpurun_out/r03_stage_classes_C1_bf16.txt 2>&1
bash tools/jobs/job70.sh
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
