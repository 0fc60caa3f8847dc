cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
for M in segformer unetpp; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$M -- python $R/bench.py --model $M --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/$M.log 2>&1
  find $O/$M -name "*kernel_stats.csv" -exec cp {} $O/${M}_kernel_stats.csv \;
  find $O/$M -name "*.csv" -size +4M -delete
  tail -1 $O/$M.log | cut -c1-300
done
