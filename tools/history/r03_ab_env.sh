#!/bin/bash
# same-box A/B of one tuning environment variable on the default bench line:  [BENCH_ARGS='--batch 4'] tools/r03_ab_env.sh VAR valueA valueB [outdir]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${4:-r03ab}; mkdir -p $O; cd $R
for rep in 1 2; do for v in $2 $3; do
  env $1=$v timeout 600 python bench.py $BENCH_ARGS --no-cpu-baseline --no-extras --no-kernel-timer 2>/dev/null | tail -1 > $O/bench_$1_${v}_$rep.json
  python - <<PY
import json
d=json.loads(open("$O/bench_$1_${v}_$rep.json").read())
print("$1=$v rep $rep: train %.1f tiles/s (%.2f ms), inference %.1f tiles/s (%.2f ms)" % (d["value"], d["ms_per_step"], d["inference_tiles_per_s"], d["inference_ms_per_step"]))
PY
done; done
