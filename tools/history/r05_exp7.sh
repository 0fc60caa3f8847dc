#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "wgrad" > $O/pytest_wgrad.txt 2>&1; grep -E "passed|failed|^FAILED" $O/pytest_wgrad.txt | tail -5 | tee -a $O/summary.txt
timeout 600 python tools/bench_wgrad.py 32 xcd > $O/bench_wgrad_xcd.txt 2>&1; grep -v amdgpu.ids $O/bench_wgrad_xcd.txt | tee -a $O/summary.txt
bash tools/r04_ab.sh r05h/ab "" "GDL_WGRAD_ROWS_XCD=1"
cat $O/ab/summary.txt >> $O/summary.txt
