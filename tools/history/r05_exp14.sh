#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/bench_loss_tail.py > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_loss_tail.csv \;
rm -rf $O/prof
python $R/tools/kernel_stats_summary.py $O/kernel_stats_loss_tail.csv | head -16 | cut -c1-150
