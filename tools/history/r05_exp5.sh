#!/bin/bash
# Round 5, experiment 5: where do eager-DDP and captured-DDP training part (per-step losses), then the FULL GPU suite (no -x).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
timeout 300 python tools/debug/r05_ddp_graph_losses.py > $O/ddp_losses.txt 2>&1; grep -E "graphed steps|eager .* graphed" $O/ddp_losses.txt | tee -a $O/summary.txt
timeout 300 python tools/debug/r05_ddp_graph_losses.py noddp > $O/noddp_losses.txt 2>&1; grep -E "graphed steps|<--" $O/noddp_losses.txt | tee -a $O/summary.txt
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_full.txt | tail -20 | tee -a $O/summary.txt
