#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh [repetitions]  (bench.py, train + inference, no extras)
# A = geo-deep-learning_amd/gdlhip/libgdlhip_ab.so (csrc: make BUILD=build_ab OUT=../gdlhip/libgdlhip_ab.so EXTRA=-D...), B = the product build
REP=${1:-2}
AB=$GRAFT_REPO_ROOT/geo-deep-learning_amd/gdlhip/libgdlhip_ab.so
for r in $(seq $REP); do
  for lib in $AB ""; do
    echo "== ${lib:-product build}"
    env GDL_LIB_PATH=$lib python bench.py --steps 40 --warmup 8 --no-extras --no-cpu-baseline --no-input-stage 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('inference_tiles_per_s'), d['ms_per_step'])"
  done
done
