cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_r03.sh 2>&1 | tail -40
python tools/stage_classes.py C1 bf16 > gpurun_out/r03_stage_classes_C1_bf16.txt 2>&1
