#!/bin/bash
# round 4, first GPU pass of the dual-resident GEMM tile: parity tests, micro-benchmark vs the 256^2 tile, block timelines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04a}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "dual_resident or epilogue_row_segments or conv or linear or gemm" 2>&1 | tail -15 | cut -c1-400 | tee $O/pytest.txt
timeout 600 python tools/bench_dual.py 32 0 > $O/bench_dual.txt 2>&1; cut -c1-400 $O/bench_dual.txt
timeout 600 python tools/probe_gemm_timeline.py 32 dual 2>&1 | cut -c1-420 | tee $O/timeline_dual.txt
