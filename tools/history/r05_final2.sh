#!/bin/bash
# Second half of the end-of-round validation: the PMC traffic passes (the first attempt traced the new `sustained` steps too and
# timed out), then the default bench line with roofline.traffic from them, and a kernel trace of the batch-4 step replayed from a
# hipGraph (how much of its 8.6 ms is kernel time?).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
bash $R/tools/pmc_bench_traffic.sh > $O/pmc.log 2>&1; tail -60 $O/pmc.log > $O/pmc_summary.txt
cp $R/gpurun_out/pmc_traffic/pmc_dominant_kernel_traffic.json $O/ && cp $O/pmc_dominant_kernel_traffic.json $R/profiles/
head -12 $O/pmc_summary.txt | cut -c1-160
T0=$(date +%s)
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
echo "default bench.py wall clock: $(( $(date +%s) - T0 )) s" | tee $O/bench_wall_clock.txt
cp $R/gpurun_out/bench_details.json $O/bench_details.json
cut -c1-1200 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b4 -- python $R/tools/bench_small_batch.py 4 30 > $O/prof_b4.log 2>&1
find $O/prof_b4 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_b4_eager_and_graph.csv \;
rm -rf $O/prof_b4
tail -2 $O/prof_b4.log | cut -c1-600
python $R/tools/kernel_stats_summary.py $O/kernel_stats_b4_eager_and_graph.csv | head -12
