#!/bin/bash
# rocprofv3 kernel statistics of the DOFA training step at the reference's batch (4 per GPU)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03z}; mkdir -p $O
B=${2:-4}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b4 -- python $R/bench.py --batch $B --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage > $O/prof_b4.log 2>&1
find $O/prof_b4 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_train_b$B.csv \;
rm -rf $O/prof_b4
tail -1 $O/prof_b4.log | cut -c1-300
python $R/tools/kernel_stats_summary.py $O/kernel_stats_train_b$B.csv | head -60
