#!/bin/bash
# Round 5, experiment 2 (one GPU call): failed-capture behaviour (3 modes, own processes), the any-ratio resized-conv kernels,
# configs[3]/[4] without fallbacks, DDP whole-step capture on a one-rank RCCL group (test + bench --force-ddp --batch 4),
# DOFA-large bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
for m in empty midway item; do
  timeout 300 python tools/debug/r05_failed_capture.py $m > $O/failed_capture_$m.txt 2>&1; echo "failed_capture $m rc=$?" | tee -a $O/summary.txt
  grep -v "^  File\|^Extension\|amdgpu.ids" $O/failed_capture_$m.txt | tail -8 | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -k "any_ratio or non_integer or fwd_sum or bwd_gather" > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_hip_model.py -q -x -k "multiband" > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_hip_tasks.py -q -x -k "multiband" > $O/pytest_tasks_multiband.txt 2>&1; tail -3 $O/pytest_tasks_multiband.txt | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_hip_tasks.py -q -x -k "ddp_training_step_captured" > $O/pytest_ddp_graph.txt 2>&1; tail -15 $O/pytest_ddp_graph.txt | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_hip_tasks.py -q -x -k "failed_graph or graph_step_auto or graphed_train" > $O/pytest_graph.txt 2>&1; tail -15 $O/pytest_graph.txt | tee -a $O/summary.txt
timeout 600 python bench.py --force-ddp --batch 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-input-stage 2>$O/bench_ddp_b4.err | tail -1 > $O/bench_ddp_b4.json
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.load(open("$O/../bench_details.json"))
    print("force-ddp batch 4: eager", d["value"], "tiles/s; ddp:", json.dumps(d.get("ddp", {}).get("graphed")))
except Exception as e: print("bench ddp b4 failed", e)
PY
cp $R/gpurun_out/bench_details.json $O/bench_ddp_b4_details.json 2>/dev/null
timeout 900 python bench.py --model dofa_large --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timer --no-input-stage 2>$O/bench_large.err | tail -1 > $O/bench_large.json
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("$O/bench_large.json").read())
    print("dofa_large b8: train", d["value"], "infer", d.get("inference_tiles_per_s"))
except Exception as e: print("bench large failed", e)
PY
grep -c UNFUSED $O/bench_large.err | sed 's/^/UNFUSED warnings in dofa_large bench: /' | tee -a $O/summary.txt
