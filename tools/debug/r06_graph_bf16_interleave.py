#!/usr/bin/env python
"""Losses of a bf16 training sequence on the tiny DOFA task: all eager vs hipGraph replays with eager steps in between
(tests/test_hip_tasks.py::test_graphed_bf16_step_with_eager_steps_in_between...).  GDL_REPACK_FUSION=0/1 as an A/B."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd" / "geo_deep_learning"))
import test_hip_tasks as T  # noqa: E402
from gdlhip import nn as gnn  # noqa: E402
from gdlhip.graphs import GraphedTrainStep  # noqa: E402


def make(capturable):
    _, task = T._dofa_task(freeze=("encoder",))
    task.trainer = T._Trainer(True)
    for blk in task.model.encoder.blocks:
        blk.drop_prob = 0.0
    task.model.aux_head.dropout_ratio = 0.0
    params = [p for p in task.parameters() if p.requires_grad]
    return task, gnn.FusedAdam(params, lr=1e-3, max_grad_norm=1.0, capturable=capturable)


def eager_step(task, opt, b):
    task.train()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = task.training_step(b, 0)
    loss.backward()
    opt.step()
    return loss


batches = [T._to_dev(T.synthetic_batch(2, 3, 112, 5, 50 + i)) for i in range(8)]
for b in batches:
    b["mask"] = b["mask"].long()
te, oe = make(False)
t2, o2 = make(False)
tg, og = make(True)
graphed = GraphedTrainStep(tg, og, batches[0], autocast_dtype=torch.bfloat16, warmup=2)
for _ in range(2):
    eager_step(te, oe, batches[0])
    eager_step(t2, o2, batches[0])
pattern = sys.argv[1] if len(sys.argv) > 1 else "ggegeggg"
for i, b in enumerate(batches):
    le = eager_step(te, oe, b).item()
    l2 = eager_step(t2, o2, b).item()
    lg = (eager_step(tg, og, b) if pattern[i] == "e" else graphed(b)).item()
    print(f"step {i} {pattern[i]}: eager {le:.6f}  second eager twin {l2:.6f} ({abs(l2 - le):.1e})  graph/eager mix {lg:.6f} ({abs(lg - le):.1e})")
