#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03q
mkdir -p $O
cd $R
timeout 900 python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03q/bench.json').read())
print(d['value'], d['ms_per_step'], d['inference_tiles_per_s'])
print(json.dumps(d.get('step_roofline'), indent=1)[:3000])
print(json.dumps(d.get('by_batch'), indent=1)[:1500])
print(json.dumps(d.get('other_models'), indent=1)[:6000])
PY
tail -3 $O/bench.err
