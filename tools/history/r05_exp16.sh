#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
GDL_LOWRES_DICE=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage > $O/prof_$v.log 2>&1
find $O/prof_$v -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train_lowres_dice_$v.csv \;
rm -rf $O/prof_$v
echo "GDL_LOWRES_DICE=$v"; python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_train_lowres_dice_$v.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("  kernel time per step (7 steps traced): %.3f ms" % (tot/1e6/7))
for r in rows:
    n=r['Name']
    if any(s in n for s in ('dice','upsample_logits','head_1x1')):
        print("  %6.1f us x %5.1f/step  %s" % (float(r['AverageNs'])/1e3, int(r['Calls'])/7, n[:90]))
PY
done
