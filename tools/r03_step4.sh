#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03ag}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_model.py -m gpu -x -q -k "bilinear or avgpool or dice or head or model or psp or upernet" 2>&1 | tail -5 | cut -c1-300 > $O/pytest.txt
cat $O/pytest.txt
for b in 32 4; do
timeout 600 python bench.py --batch $b --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>/dev/null | tail -1 > $O/bench_b$b.json
python - <<PY
import json
d=json.loads(open("$O/bench_b$b.json").read())
print("batch $b: train %.1f tiles/s (%.2f ms), inference %.1f tiles/s (%.2f ms)" % (d["value"], d["ms_per_step"], d["inference_tiles_per_s"], d["inference_ms_per_step"]))
PY
done
bash tools/r03_profile_b4.sh $1 4 | tail -62 > $O/prof_b4_summary.txt
