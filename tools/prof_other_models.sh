#!/bin/bash
# round 6: kernel traces of the two other models of the path (SegFormer-B2, UNet++ / ResNet18) at batch 32, training only
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_other_models
rm -rf $OUT; mkdir -p $OUT
for m in segformer unetpp; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -- python $GRAFT_REPO_ROOT/bench.py --model $m --batch 32 --mode train --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-input-stage --no-kernel-timer > $OUT/$m.log 2>&1
  f=$(ls $OUT/$m/*/*kernel_stats.csv | head -1)
  cp $f $OUT/kernel_stats_${m}_train_b32.csv
  python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1].split("/")[-3], "%.1f ms of kernel time" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:6.2f}% {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
P
  find $OUT/$m -type f -size +4M -delete
done
