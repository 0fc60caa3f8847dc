#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "resize or up4 or resized or concat or tap or epilogue or batchnorm" 2>&1 | tail -15 > $O/pytest_ops.txt
cat $O/pytest_ops.txt
timeout 300 python tools/bench_tapsum.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_tapsum.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-kernel-timer 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03l/bench.json').read())
print(d['value'], d['ms_per_step'], d['inference_tiles_per_s'], d['inference_ms_per_step'])
PY
