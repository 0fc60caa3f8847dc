#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "epilogue or linear or conv_nhwc or shared_staging" 2>&1 | tail -15 > $O/pytest_ops.txt
cat $O/pytest_ops.txt
timeout 300 python tools/probe_gemm_timeline.py 32 epilogue 2>&1 | grep -v amdgpu.ids | tee $O/gemm_timeline_epilogue_ab.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras 2>$O/bench.err | tail -1 > $O/bench.json
cut -c1-250 $O/bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench.json').read())
print(d['value'], d['inference_tiles_per_s'], d['roofline']['by_k_depth'])
PY
