#!/bin/bash
# classifier head, second pass (wide weight-gradient partials): parity, per-kernel micro-benchmark, in-step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "head or dice" > $O/pytest_ops.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_ops.txt | tail -6
for v in 0 1; do echo "GDL_HEAD_MFMA=$v"; GDL_HEAD_MFMA=$v timeout 300 python tools/bench_loss_tail.py 2>&1 | grep "head 1x1"; done | tee $O/bench_head.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/bench_loss_tail.py > $O/prof.log 2>&1 )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_loss_tail.csv \; ; rm -rf $O/prof
python - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats_bench_loss_tail.csv")):
    if "head_1x1" in r["Name"]: print("%8.1f us (min %8.1f) x %4s  %s" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, r["Calls"], r["Name"][:80]))
PY
bash tools/r04_ab.sh r05t/ab "GDL_HEAD_MFMA=0" "GDL_HEAD_MFMA=1"
timeout 900 python -m pytest tests/test_hip_tasks.py tests/test_hip_model.py -q -k "dofa or graph or ddp or trainer or tiny or base_512" > $O/pytest.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -6
