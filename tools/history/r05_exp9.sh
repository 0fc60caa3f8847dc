#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R
timeout 600 python tools/bench_input_stage.py 60 2>&1 | grep -v amdgpu.ids | tee $O/bench_input_stage.txt
