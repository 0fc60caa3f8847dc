#!/bin/bash
# same-box A/B of tuning environment variables on the default bench line (train + inference, no extras):
#   tools/r04_ab.sh outdir "VAR=a VAR2=b" "VAR=c" ...    (each quoted argument is one arm; "" = defaults); two interleaved repetitions
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for rep in 1 2; do i=0; for arm in "$@"; do i=$((i+1))
  env $arm timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>/dev/null | tail -1 > $O/arm${i}_$rep.json
  python - <<PY
import json
d=json.loads(open("$O/arm${i}_$rep.json").read())
print("arm $i [$arm] rep $rep: train %.1f tiles/s (%.2f ms), inference %.1f tiles/s (%.2f ms)" % (d["value"], d["ms_per_step"], d["inference_tiles_per_s"], d["inference_ms_per_step"]))
PY
done; done 2>&1 | tee $O/summary.txt
