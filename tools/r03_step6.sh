#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_unetpp.py -x -q -m gpu -s -k "production_shape or 512_bf16" 2>&1 | grep -v "^$" | tail -45 > $O/pytest.txt
cat $O/pytest.txt
