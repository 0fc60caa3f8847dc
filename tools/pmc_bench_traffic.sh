#!/bin/bash
# HBM traffic of the kernels of one bench.py training step: separate --pmc passes with --kernel-trace only, as
# MI355X_MICROARCH.md "HBM" prescribes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass; on gfx950
# FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled by tools/pmc_bench_traffic.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_traffic
mkdir -p $O
CMD="python $R/bench.py --batch ${GDL_PMC_BATCH:-64} --steps 1 --warmup 1 --mode train --no-cpu-baseline --no-kernel-timer --no-extras --no-input-stage --min-seconds 0"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pass_$i -- $CMD > $O/pass_$i.log 2>&1
done
python $R/tools/pmc_bench_traffic.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +8M -delete
cat $O/summary.txt
