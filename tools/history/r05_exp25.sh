#!/bin/bash
# after the wide / scaled head kernels: DOFA task + model tests (the auxiliary head's backward now takes the scaled kernels), smoke, headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O; cd $R
timeout 130 python -m pytest tests/test_hip_tasks.py tests/test_hip_model.py -q -k "dofa or graph or ddp or trainer or tiny or base_512 or syncbn or multiband" > $O/pytest.txt 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest.txt | tail -6
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>/dev/null | tail -1 > $O/dofa.json
python -c "
import json; d=json.loads(open('$O/dofa.json').read()); print('DOFA-base b32: train %.1f tiles/s (%.2f ms), inference %.1f (%.2f ms)' % (d['value'], d['ms_per_step'], d['inference_tiles_per_s'], d['inference_ms_per_step']))" | tee $O/dofa.txt
