#!/bin/bash
# End-of-round validation on the GPU box: full -m gpu suite, smoke(), the default bench line, rocprofv3 kernel statistics of
# the same command and the per-kernel PMC traffic passes.  Everything lands under gpurun_out/r02z.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02z
mkdir -p $O
cd $R
case " $* " in *" notest "*) echo "(pytest skipped)" > $O/pytest_gpu.txt ;; *)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/pytest_gpu.txt ;; esac
cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
cat $O/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*.csv" -size +4M -delete
head -12 $O/kernel_stats.csv
case " $* " in *" pmc "*) bash $R/tools/pmc_bench_traffic.sh > $O/pmc.log 2>&1 ;; esac
tail -30 $O/pmc.log
