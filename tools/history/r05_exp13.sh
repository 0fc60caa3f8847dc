#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_hip_ops.py -q -k "head_and_logit or dice" 2>&1 | tail -2 | tee $O/pytest.txt
timeout 300 python tools/bench_loss_tail.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_loss_tail.txt
