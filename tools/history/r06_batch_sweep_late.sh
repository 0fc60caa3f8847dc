#!/bin/bash
# per-GPU batch 64 / 96 / 128 on one box at the round's final kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R; mkdir -p gpurun_out/r06v2
for rep in 1 2; do for b in 64 96 128; do
  echo "batch $b: $(python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage --min-seconds 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('inference_tiles_per_s'), d['ms_per_step'])")"
done; done | tee gpurun_out/r06v2/batch_sweep_late.txt
