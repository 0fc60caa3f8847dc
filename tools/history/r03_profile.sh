#!/bin/bash
# rocprofv3 kernel statistics of the DOFA training step and of the inference step (batch 32), separately
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03p}; mkdir -p $O
for mode in train infer; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -- python $R/bench.py --mode $mode --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timer --min-seconds 0 > $O/prof_$mode.log 2>&1
  find $O/prof_$mode -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$mode.csv \;
  rm -rf $O/prof_$mode
  python $R/tools/kernel_stats_summary.py $O/kernel_stats_$mode.csv | head -45
done
