#!/bin/bash
# HBM traffic of the kernels of one bench.py training step (separate --pmc passes, kernel-trace only; see
# MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled by tools/pmc_traffic_summary.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --batch 32 --steps 1 --warmup 1 --mode train --no-cpu-baseline --no-kernel-timer"
timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_1 -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_2 -- $CMD > /dev/null 2>&1
ls $R/gpurun_out/pmc_traffic_1/*/ $R/gpurun_out/pmc_traffic_2/*/ | head
