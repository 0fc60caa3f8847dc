#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
timeout 300 python tools/debug/r05_ddp_graph_losses.py > $O/ddp_losses.txt 2>&1; grep -E "graphed steps|eager .* graphed|Error|error" $O/ddp_losses.txt | cut -c1-120 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_hip_tasks.py -q -k "ddp or graph" > $O/pytest_tasks.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_tasks.txt | tail -8 | tee -a $O/summary.txt
timeout 600 python bench.py --force-ddp --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --with-input-stage 2>$O/bench_ddp_b4.err | tail -1 > $O/bench_ddp_b4.json; cut -c1-1500 $O/bench_ddp_b4.json | tee -a $O/summary.txt
cp $R/gpurun_out/bench_details.json $O/bench_ddp_b4_details.json
