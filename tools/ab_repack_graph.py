#!/usr/bin/env python
"""Training / inference tiles/s of DOFA-base + UperNet at a small per-GPU batch from a hipGraph replay (the trainer's default there)
and launched eagerly -- one process per configuration, so environment switches (GDL_REPACK_FUSION, GDL_CONV_STAGE4, ...) apply.
usage: ab_repack_graph.py [batch] [steps]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.side_measurement("dofa", b, steps, 10, dev, True, graphs=True)
print(json.dumps({"batch": b, "graph_train": r.get("train_tiles_per_s"), "graph_infer": r.get("inference_tiles_per_s"),
                  "eager_train": r.get("eager", {}).get("train_tiles_per_s"), "eager_infer": r.get("eager", {}).get("inference_tiles_per_s"),
                  "error": r.get("hipgraph", {}).get("error")}))
