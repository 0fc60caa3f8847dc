#!/bin/bash
# rocprofv3 kernel trace (every launch with its start / end) and statistics of the eager training step at the reference's own
# per-GPU batch (configs/dofa_config_RGB.yaml:85).   RUN=r06q tools/prof_batch4.sh   -> gpurun_out/$RUN/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; RUN=${RUN:-r06q}; O=$R/gpurun_out/$RUN; mkdir -p $O; cd $R
B=${B:-4}
for MODE in train infer; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$MODE -- python $R/bench.py --batch $B --mode $MODE --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage --min-seconds 0 > $O/prof_$MODE.log 2>&1 )
  find $O/prof_$MODE -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_dofa_${MODE}_only_b$B.csv \;
  find $O/prof_$MODE -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_dofa_${MODE}_only_b$B.csv \;
  rm -rf $O/prof_$MODE
  tail -2 $O/prof_$MODE.log
done
python tools/kernel_stats_summary.py $O/kernel_stats_dofa_train_only_b$B.csv | head -40
