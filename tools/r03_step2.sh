#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "resize or up4 or resized or concat or tap" 2>&1 | tail -15 > $O/pytest_ops.txt
cat $O/pytest_ops.txt
timeout 300 python tools/bench_tapsum.py 2>&1 | tee $O/bench_tapsum.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras 2>$O/bench.err | tail -1 > $O/bench.json
cut -c1-900 $O/bench.json
