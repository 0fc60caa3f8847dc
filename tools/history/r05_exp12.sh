#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "dice or head_and_logit" > $O/pytest_ops.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_ops.txt | tail -6 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_hip_tasks.py -q -k "dofa or graph or ddp or trainer" > $O/pytest_tasks.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_tasks.txt | tail -6 | tee -a $O/summary.txt
bash tools/r04_ab.sh r05m/ab "GDL_LOWRES_DICE=0" "GDL_LOWRES_DICE=1"
cat $O/ab/summary.txt >> $O/summary.txt
