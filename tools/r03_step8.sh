#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_tasks.py -x -q -m gpu -k "graphed" 2>&1 | grep -v "UserWarning\|^$" | tail -8 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python -X faulthandler - > $O/graph_dbg.txt 2>&1 <<'PY'
import sys, json, torch, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/geo-deep-learning_amd")
import bench
from gdlhip.graphs import GraphedEvalStep, GraphedTrainStep
dev = torch.device("cuda", 0)
task, optimizer = bench.build_task("dofa", dev, False, 0, capturable=True)
batch = bench.synthetic_batch(4, dev, 43)
print("built", flush=True)
gt = GraphedTrainStep(task, optimizer, batch, autocast_dtype=torch.bfloat16)
print("captured train", flush=True)
for _ in range(3):
    l = gt()
torch.cuda.synchronize()
print("replayed", float(l), flush=True)
dt = bench.timed(lambda: gt(), 10, 2, 1, dev)
print("graph train ms/step", 1e3 * dt / 10, flush=True)
task.eval()
ge = GraphedEvalStep(lambda b: task.validation_step(b, 0), batch, autocast_dtype=torch.bfloat16)
print("captured eval", flush=True)
dt = bench.timed(lambda: ge(), 10, 2, 1, dev)
print("graph eval ms/step", 1e3 * dt / 10, flush=True)
PY
tail -40 $O/graph_dbg.txt | cut -c1-300
