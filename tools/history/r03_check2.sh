#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03chk2}; mkdir -p $O; cd $R
timeout 300 python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/stem.txt
for m in unetpp segformer; do
timeout 600 python bench.py --model $m --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>/dev/null | tail -1 > $O/bench_$m.json
python - <<PY
import json
d=json.loads(open("$O/bench_$m.json").read())
print("$m: train %.1f tiles/s (%.2f ms), inference %.1f tiles/s (%.2f ms)" % (d["value"], d["ms_per_step"], d["inference_tiles_per_s"], d["inference_ms_per_step"]))
PY
done
