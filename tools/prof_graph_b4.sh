#!/bin/bash
# rocprofv3 kernel statistics of hipGraph replays of the training step at a small batch (tools/ab_repack_graph.py).  RUN=r06q B=4 tools/prof_graph_b4.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; RUN=${RUN:-r06q}; O=$R/gpurun_out/$RUN; mkdir -p $O; cd $R
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python $R/tools/ab_repack_graph.py ${B:-4} 50 > $O/prof_graph.log 2>&1 )
find $O/prof_graph -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_graph_b${B:-4}${TAG}.csv \;
rm -rf $O/prof_graph; tail -1 $O/prof_graph.log
python tools/kernel_stats_summary.py $O/kernel_stats_graph_b${B:-4}${TAG}.csv cast_kernel pack_dgrad elementwise multi_repack multi_adam flip Fill
