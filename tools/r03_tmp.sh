cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03r
timeout 600 python -X faulthandler - > gpurun_out/r03r/graph_dbg2.txt 2>&1 <<'PY'
import sys, json, torch, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/geo-deep-learning_amd")
import bench
dev = torch.device("cuda", 0)
r = bench.side_measurement("dofa", 4, 10, 3, dev, True, graphs=True)
print(4, json.dumps(r), flush=True)
PY
tail -50 gpurun_out/r03r/graph_dbg2.txt | cut -c1-250
