#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_tasks.py tests/test_input_stage.py tests/test_datamodule.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-kernel-timer --mode train 2>$O/bench.err | tail -1 > $O/bench_plain.json
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-kernel-timer --mode train --force-ddp 2>$O/bench_ddp.err | tail -1 > $O/bench_ddp.json
python - <<'PY'
import json
for f in ("bench_plain","bench_ddp"):
    d=json.loads(open(f"gpurun_out/r03m/{f}.json").read())
    print(f, d["value"], d["ms_per_step"], d.get("pcie_inclusive"), d.get("ddp"))
PY
tail -5 $O/bench_ddp.err
