#!/bin/bash
# End-of-round validation on the GPU box: full -m gpu suite, smoke(), the default bench line, rocprofv3 kernel statistics of
# the training step and of the inference step, the per-layer conv / wgrad tables of the three models, and the per-kernel PMC
# traffic passes.  Everything lands under gpurun_out/r03z.      tools/r03_final.sh [notest] [pmc] [plans]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
case " $* " in *" notest "*) echo "(pytest skipped)" > $O/pytest_gpu.txt ;; *)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/pytest_gpu.txt ;; esac
cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1200 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
cat $O/bench.json | cut -c1-400
bash $R/tools/r03_profile.sh r03z > $O/profile_summary.txt 2>&1
case " $* " in *" plans "*)
for m in dofa segformer unetpp; do timeout 300 python tools/log_conv_plans.py $m 32 > $O/conv_plans_$m.txt 2>&1; done ;; esac
case " $* " in *" pmc "*) bash $R/tools/pmc_bench_traffic.sh > $O/pmc.log 2>&1; tail -45 $O/pmc.log > $O/pmc_summary.txt ;; esac
ls $O
