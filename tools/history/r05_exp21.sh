#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
timeout 400 python tools/debug/r05_small_kernel_sources.py 32 > $O/small_kernel_sources_b32.txt 2>$O/err.txt; tail -60 $O/small_kernel_sources_b32.txt; tail -5 $O/err.txt
