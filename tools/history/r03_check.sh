#!/bin/bash
# quick re-validation after a host-side change: optimizer / graph / BatchNorm tests + batch 2 / 4 eager-vs-graph rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03chk}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_tasks.py -m gpu -x -q -k "adam or graphed or batchnorm or task_steps" 2>&1 | tail -3 | cut -c1-300 | tee $O/pytest.txt
timeout 600 python tools/bench_bn.py > $O/bn.txt 2>&1; cut -c1-260 $O/bn.txt
python - <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-kernel-timer", "--no-input-stage"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
d = json.loads(out)
open("gpurun_out/%s/bench.json" % sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03chk/bench.json", "w").write(out)
print("train %.1f inference %.1f" % (d["value"], d["inference_tiles_per_s"]))
for b, v in d["by_batch"].items():
    print("batch", b, "eager %.1f / %.1f" % (v["train_tiles_per_s"], v["inference_tiles_per_s"]), "graph %.1f / %.1f" % (v["hipgraph"]["train_tiles_per_s"], v["hipgraph"]["inference_tiles_per_s"]))
PY
