#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03ad}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_transformer_bwd.py tests/test_hip_model.py tests/test_hip_segformer.py -m gpu -x -q -k "attention or flash or block or model or eval or train" 2>&1 | tail -8 | cut -c1-300 > $O/pytest.txt
cat $O/pytest.txt
python - <<'PY'
import ctypes, sys
sys.path.insert(0, "geo-deep-learning_amd")
PY
for v in 2 3; do
GDL_FLASH_FWD=$v timeout 600 python bench.py --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>/dev/null | tail -1 > $O/bench_flash$v.json
python - <<PY
import json
d=json.loads(open("$O/bench_flash$v.json").read())
print("GDL_FLASH_FWD=$v: train %.1f tiles/s (%.2f ms), inference %.1f tiles/s (%.2f ms)" % (d["value"], d["ms_per_step"], d["inference_tiles_per_s"], d["inference_ms_per_step"]))
PY
done
