#!/bin/bash
# BatchNorm final reductions on 1024 threads + feature-gradient grid: parity, per-kernel times from a train-only trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -q -k "bn or batchnorm or norm or head or stats or conv_bn or sync" > $O/pytest_ops.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_ops.txt | tail -6
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage > $O/prof.log 2>&1 )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train.csv \; ; rm -rf $O/prof
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_train.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step: %.3f ms" % (tot/1e6/7))
for r in rows:
    if any(s in r["Name"] for s in ("head_1x1","bn_stats_final","bn_bwd_final","dice")): print("%8.1f us x %5.1f/step  %s" % (float(r["AverageNs"])/1e3, int(r["Calls"])/7, r["Name"][:80]))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.1f tiles/s (%.2f ms), inference %.1f (%.2f ms)' % (d['value'], d['ms_per_step'], d['inference_tiles_per_s'], d['inference_ms_per_step']))"
timeout 900 python -m pytest tests/test_hip_tasks.py tests/test_hip_model.py -q -k "dofa or graph or ddp or trainer or tiny or base_512 or syncbn" > $O/pytest.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -6
