#!/bin/bash
# End-of-round validation on the GPU box (round 5): PMC traffic passes of the final GEMM sources, full -m gpu suite, smoke(), the
# default bench line with its wall-clock time, rocprofv3 kernel statistics of the training and the inference step, the per-layer
# conv / wgrad tables, the DDP (one-rank RCCL) line at batch 4.  Everything lands under gpurun_out/r05z.
#      tools/r05_final.sh [notest] [nopmc]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN:-r05z}
mkdir -p $O
cd $R
case " $* " in *" nopmc "*) ;; *) bash $R/tools/pmc_bench_traffic.sh > $O/pmc.log 2>&1; tail -60 $O/pmc.log > $O/pmc_summary.txt
  cp $R/gpurun_out/pmc_traffic/pmc_dominant_kernel_traffic.json $O/ && cp $O/pmc_dominant_kernel_traffic.json $R/profiles/ ;; esac   # the bench below then reports roofline.traffic from these passes
case " $* " in *" notest "*) echo "(pytest skipped)" > $O/pytest_gpu.txt ;; *)
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu_full.txt | tail -8 > $O/pytest_gpu.txt ;; esac
cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
T0=$(date +%s)
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
echo "default bench.py wall clock: $(( $(date +%s) - T0 )) s" | tee $O/bench_wall_clock.txt
cp $R/gpurun_out/bench_details.json $O/bench_details.json
cat $O/bench.json | cut -c1-3200
T0=$(date +%s)
timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_driver_flags.err | tail -1 > $O/bench_driver_flags.json
echo "bench.py --steps 20 --warmup 5 (the driver's flags) wall clock: $(( $(date +%s) - T0 )) s" | tee -a $O/bench_wall_clock.txt
bash $R/tools/r03_profile.sh ${RUN:-r05z} > $O/profile_summary.txt 2>&1
timeout 600 python bench.py --force-ddp --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-input-stage 2>$O/bench_ddp_b4.err | tail -1 > $O/bench_ddp_b4.json
cp $R/gpurun_out/bench_details.json $O/bench_ddp_b4_details.json
for m in dofa; do timeout 300 python tools/log_conv_plans.py $m 32 > $O/conv_plans_$m.txt 2>&1; done
ls $O
