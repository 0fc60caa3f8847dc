#!/bin/bash
# Round 5, experiment 1 (one GPU call): parity of the new kernels, attention forward schedules, same-box A/B of the pipelined
# attention forward (GDL_FLASH_FWD=4/5) and of the BatchNorm-backward-fused gather (GDL_FUSE_BN_BWD_GATHER).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "flash or gather or resized_conv or conv_bn" 2>&1 | tail -5 > $O/pytest_ops.txt
cat $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_hip_tasks.py -q -x -k "graph or failed" 2>&1 | tail -5 > $O/pytest_tasks.txt
cat $O/pytest_tasks.txt
timeout 600 python -m pytest tests/test_hip_model.py -q -x -k "tiny or base_512" 2>&1 | tail -5 > $O/pytest_model.txt
cat $O/pytest_model.txt
timeout 300 python tools/bench_attention.py > $O/bench_attention.txt 2>&1; grep -E "forward schedules|fwd v3" $O/bench_attention.txt
bash tools/r04_ab.sh r05a/ab "" "GDL_FLASH_FWD=4" "GDL_FLASH_FWD=5" "GDL_FUSE_BN_BWD_GATHER=0" "GDL_FLASH_FWD=4 GDL_FUSE_BN_BWD_GATHER=0"
