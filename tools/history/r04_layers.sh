#!/bin/bash
# per-variant / per-layer TF/s of the train step's conv_gemm launches under an environment setting:  tools/r04_layers.sh "VAR=x ..."
R=$GRAFT_REPO_ROOT; cd $R
env $1 timeout 600 python bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline --no-extras --no-input-stage 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$1] train %.1f tiles/s' % d['value'])
print('  dominant:', r['kernel'][:60], '%.0f TF/s share %.3f' % (r['achieved'], r['share_of_step_time']))
for l in r['by_layer'][:8]: print('     ', l)
for k,v in r['other_conv_gemm_variants'].items(): print('  other:', k[:60], v)
"
