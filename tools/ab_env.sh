#!/bin/bash
# same-box A/B through an environment switch: tools/r06_ab.sh VAR OFF ON [repetitions]  (bench.py, train + inference, no extras)
VAR=$1; OFF=$2; ON=$3; REP=${4:-2}
for r in $(seq $REP); do
  for val in $OFF $ON; do
    echo "== $VAR=$val"
    env $VAR=$val python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-input-stage 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('inference_tiles_per_s'), d['ms_per_step'])"
  done
done
