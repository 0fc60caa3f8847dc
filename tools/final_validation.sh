#!/bin/bash
# End-of-round validation on ONE box: full -m gpu suite, smoke(), the default bench line (wall clock), the driver's flags, the
# rocprofv3 kernel statistics of the training and of the inference step, the PMC passes behind roofline.traffic.
#   RUN=r06w tools/final_validation.sh [notest] [nopmc]      -> gpurun_out/$RUN/*  (copy what is to be judged into profiles/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; RUN=${RUN:-r06w}; O=$R/gpurun_out/$RUN; mkdir -p $O; cd $R
if [[ " $* " != *" notest "* ]]; then
  timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.txt 2>&1
  grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu_full.txt | tail -8 | tee $O/pytest_gpu.txt
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
if [[ " $* " != *" nopmc "* ]]; then      # first: bench.py reads the refreshed profiles/pmc_dominant_kernel_traffic.json
  timeout 1500 bash tools/pmc_bench_traffic.sh > $O/pmc_summary.txt 2>&1
  cp gpurun_out/pmc_traffic/summary.txt $O/pmc_per_kernel.txt 2>/dev/null
  # the summary keyed by the GEMM sources' hash: into profiles/ on this box (the bench below reads it) and into the run's
  # directory (merged back; commit it as profiles/pmc_dominant_kernel_traffic.json)
  cp gpurun_out/pmc_traffic/pmc_dominant_kernel_traffic.json profiles/pmc_dominant_kernel_traffic.json 2>/dev/null
  cp gpurun_out/pmc_traffic/pmc_dominant_kernel_traffic.json $O/pmc_dominant_kernel_traffic.json 2>/dev/null
fi
T0=$(date +%s)
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
echo "default bench.py wall clock: $(( $(date +%s) - T0 )) s" | tee $O/bench_wall_clock.txt
cp $R/gpurun_out/bench_details.json $O/bench_details.json
cut -c1-700 $O/bench.json; echo; python -c "
import json; d=json.load(open('$O/bench_details.json'))
print('train', d['value'], 'infer', d['inference_tiles_per_s'], 'pcie', d['pcie_inclusive']['train_tiles_per_s'], 'roofline', d['roofline']['kernel'][:40], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print({k:(v['train_tiles_per_s'],v['inference_tiles_per_s']) for k,v in d['other_models'].items()})
print({k:(v['train_tiles_per_s'],v['inference_tiles_per_s']) for k,v in d['by_batch'].items()})
print('step_roofline', {k:v.get('frac_of_bound') for k,v in d.get('step_roofline',{}).items()})"
T0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench_driver_flags.err | tail -1 > $O/bench_driver_flags.json
echo "bench.py --steps 20 --warmup 5 (the driver's flags) wall clock: $(( $(date +%s) - T0 )) s" | tee -a $O/bench_wall_clock.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_flags.json').read()); print('driver flags: train', d['value'], 'infer', d['inference_tiles_per_s'], 'sustained', d.get('sustained'), 'line bytes', len(json.dumps(d, separators=(',',':'))))"
for MODE in train infer; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$MODE -- python $R/bench.py --mode $MODE --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage --min-seconds 0 > $O/prof_$MODE.log 2>&1 )
  find $O/prof_$MODE -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_dofa_${MODE}_only_b64.csv \; ; rm -rf $O/prof_$MODE
done
python tools/kernel_stats_summary.py $O/kernel_stats_dofa_train_only_b64.csv 2>/dev/null | head -30 | tee $O/kernel_stats_train_summary.txt
