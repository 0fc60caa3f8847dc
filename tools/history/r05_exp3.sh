#!/bin/bash
# Round 5, experiment 3: failed-capture recovery, DDP capture on one stream, small-map BatchNorm (parity + batch-4 A/B), full suite.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
for m in empty midway item; do
  timeout 300 python tools/debug/r05_failed_capture.py $m > $O/failed_capture_$m.txt 2>&1; echo "failed_capture $m rc=$?" | tee -a $O/summary.txt
  grep -v "^  File\|^Extension\|amdgpu.ids\|Warning\|^  " $O/failed_capture_$m.txt | tail -8 | tee -a $O/summary.txt
done
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "batchnorm" > $O/pytest_bn.txt 2>&1; tail -3 $O/pytest_bn.txt | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_hip_tasks.py -q -x -k "ddp_training_step_captured" > $O/pytest_ddp_graph.txt 2>&1; grep -v "^  File \"/usr/local/lib/python3.10/dist-packages/\(_pytest\|pluggy\)" $O/pytest_ddp_graph.txt | tail -60 | tee -a $O/summary.txt
timeout 600 python bench.py --force-ddp --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-input-stage 2>$O/bench_ddp_b4.err | tail -1 > $O/bench_ddp_b4.json
grep -A40 "DDP step capture failed" $O/bench_ddp_b4.err | head -60 | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.load(open("$R/gpurun_out/bench_details.json"))
    print("force-ddp batch 4: eager", d["value"], "tiles/s; ddp.graphed:", json.dumps(d.get("ddp", {}).get("graphed")))
except Exception as e: print("bench ddp b4 failed", e)
PY
for rep in 1 2; do for px in 8192 0; do
  echo "GDL_BN_SMALL_PIXELS=$px rep $rep: $(GDL_BN_SMALL_PIXELS=$px timeout 300 python tools/bench_small_batch.py 4 30 2>/dev/null | tail -1)" | tee -a $O/summary.txt
done; done
echo "batch 2: $(timeout 300 python tools/bench_small_batch.py 2 30 2>/dev/null | tail -1)" | tee -a $O/summary.txt
echo "batch 8: $(timeout 300 python tools/bench_small_batch.py 8 20 2>/dev/null | tail -1)" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_full.txt 2>&1; tail -4 $O/pytest_full.txt | tee -a $O/summary.txt
