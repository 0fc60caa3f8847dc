#!/usr/bin/env python
"""Train / inference tiles/s of the DOFA step at one small per-GPU batch, eager and replayed from a hipGraph (bench.py's
`by_batch` entry on its own: for same-box A/B runs through environment switches).   python tools/bench_small_batch.py [batch] [steps]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "geo-deep-learning_amd")]
sys.argv, args = sys.argv[:1], sys.argv[1:]
import bench  # noqa: E402

b = int(args[0]) if args else 4
steps = int(args[1]) if len(args) > 1 else 30
torch.cuda.set_device(0)
out = bench.side_measurement("dofa", b, steps, 5, torch.device("cuda", 0), True, graphs=True)
print(json.dumps({k: out[k] for k in ("per_gpu_batch", "train_tiles_per_s", "inference_tiles_per_s", "train_ms_per_step",
                                      "inference_ms_per_step", "eager", "hipgraph") if k in out}))
