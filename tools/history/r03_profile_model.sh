#!/bin/bash
# rocprofv3 kernel statistics of one model's training step:  tools/r03_profile_model.sh outdir model [batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O
M=${2:-segformer}; B=${3:-32}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$M -- python $R/bench.py --model $M --batch $B --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage > $O/prof_$M.log 2>&1
find $O/prof_$M -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_${M}_train_b$B.csv \;
rm -rf $O/prof_$M
tail -1 $O/prof_$M.log | cut -c1-200
python $R/tools/kernel_stats_summary.py $O/kernel_stats_${M}_train_b$B.csv | head -40
