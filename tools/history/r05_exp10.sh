#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -q -s -k "batchnorm or plain_conv_modules" > $O/pytest_ops.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR|^\{" $O/pytest_ops.txt | cut -c1-900 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_hip_model.py -q -s -k "base_512_eval" > $O/pytest_model.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR|bf16 vs the f32" $O/pytest_model.txt | cut -c1-400 | tee -a $O/summary.txt
