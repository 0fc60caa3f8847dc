#!/bin/bash
# classifier head: MFMA forward / register-weight feature gradient / 2048 partial rows -- parity, micro-benchmark, in-step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "head or dice" > $O/pytest_ops.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_ops.txt | tail -6
for v in 0 1; do echo "GDL_HEAD_MFMA=$v"; GDL_HEAD_MFMA=$v timeout 300 python tools/bench_loss_tail.py 2>&1 | grep "head 1x1"; done | tee $O/bench_head.txt
bash tools/r04_ab.sh r05s/ab "GDL_HEAD_MFMA=0" "GDL_HEAD_MFMA=1"
timeout 900 python -m pytest tests/test_hip_tasks.py tests/test_hip_model.py -q -k "dofa or graph or ddp or trainer or tiny or base_512" > $O/pytest.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -6
