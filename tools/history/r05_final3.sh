#!/bin/bash
# Last validation of the round on the final code: full -m gpu suite, smoke(), the default bench line (wall clock), the driver's flags.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu_full.txt | tail -8 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
T0=$(date +%s)
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
echo "default bench.py wall clock: $(( $(date +%s) - T0 )) s" | tee $O/bench_wall_clock.txt
cp $R/gpurun_out/bench_details.json $O/bench_details.json
cut -c1-600 $O/bench.json; echo; python -c "
import json; d=json.load(open('$O/bench_details.json'))
print('train', d['value'], 'infer', d['inference_tiles_per_s'], 'pcie', d['pcie_inclusive']['train_tiles_per_s'], 'traffic', d['roofline']['traffic'], 'frac', d['roofline']['frac'])
print({k:(v['train_tiles_per_s'],v['inference_tiles_per_s']) for k,v in d['other_models'].items()})
print({k:(v['train_tiles_per_s'],v['inference_tiles_per_s']) for k,v in d['by_batch'].items()})"
T0=$(date +%s)
timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_driver_flags.err | tail -1 > $O/bench_driver_flags.json
echo "bench.py --steps 20 --warmup 5 (the driver's flags) wall clock: $(( $(date +%s) - T0 )) s" | tee -a $O/bench_wall_clock.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_flags.json').read()); print('driver flags: train', d['value'], 'infer', d['inference_tiles_per_s'], 'sustained', d.get('sustained'), 'pcie', d.get('pcie_inclusive'), 'line bytes', len(json.dumps(d, separators=(',',':'))))"
