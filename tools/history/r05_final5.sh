#!/bin/bash
# clean full -m gpu suite + smoke() on the final code (after the small-map BatchNorm test fix)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05w2; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu_full.txt | tail -8 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
