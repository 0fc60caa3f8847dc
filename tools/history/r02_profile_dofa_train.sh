cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*.csv" -size +4M -delete
python $R/tools/kernel_stats_summary.py $O/kernel_stats.csv | head -40
