#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "batchnorm_small or head_mfma or dice_loss_from_low" > $O/pytest_ops.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_ops.txt | tail -6
