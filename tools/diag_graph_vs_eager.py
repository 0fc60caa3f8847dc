#!/usr/bin/env python
"""Diagnostic for tests/test_hip_tasks.py::test_graphed_train_step_matches_eager: per-step parameter and gradient deviations
between the eager twin and the graph replay, for the parameters that deviate most."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
sys.path.insert(0, str(ROOT / "tests"))
import test_hip_tasks as T  # noqa: E402
from gdlhip import nn as gnn  # noqa: E402
from gdlhip.graphs import GraphedTrainStep  # noqa: E402
from oracle import synthetic_batch  # noqa: E402


def make(capturable):
    _, task = T._dofa_task(freeze=("encoder",))
    task.trainer = T._Trainer(True)
    for blk in task.model.encoder.blocks:
        blk.drop_prob = 0.0
    task.model.aux_head.dropout_ratio = 0.0
    params = [p for p in task.parameters() if p.requires_grad]
    return task, gnn.FusedAdam(params, lr=1e-3, max_grad_norm=1.0, capturable=capturable)


batches = [T._to_dev(synthetic_batch(2, 3, 112, 5, 30 + i)) for i in range(5)]
for b in batches:
    b["mask"] = b["mask"].long()
te, oe = make(False)
tg, og = make(True)
graphed = GraphedTrainStep(tg, og, batches[0], autocast_dtype=None, warmup=2)
te.train()
for _ in range(2):
    oe.zero_grad(set_to_none=True)
    te.training_step(batches[0], 0).backward()
    oe.step()
pe, pg = dict(te.named_parameters()), dict(tg.named_parameters())
name = "model.neck.lateral_convs.3.conv.weight"


def report(tag):
    worst = sorted(((float((pg[n] - pe[n]).abs().max()), n) for n in pg if pg[n].requires_grad), reverse=True)[:3]
    print(f"{tag}: max |param diff| {worst}", flush=True)


report("after the two warm-up steps")
for i, b in enumerate(batches):
    oe.zero_grad(set_to_none=True)
    le = te.training_step(b, 0)
    le.backward()
    ge = pe[name].grad.clone()
    oe.step()
    lg = graphed(b)
    gg = pg[name].grad
    print(f"step {i}: loss eager {le.item():.7f} graph {lg.item():.7f}; {name} grad |eager| {ge.norm().item():.3e} |graph| {gg.norm().item():.3e} "
          f"|diff| {(ge - gg).norm().item():.3e}; grad-norm all (eager) {torch.sqrt(sum((p.grad.float() ** 2).sum() for p in pe.values() if p.grad is not None)).item():.3e}")
    report(f"step {i}")
