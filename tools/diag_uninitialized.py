#!/usr/bin/env python
"""Diagnostic: does any kernel read memory it (or a predecessor) never wrote?  torch.empty is made to return NaN-filled buffers
(torch.utils.deterministic.fill_uninitialized_memory) and one training step of each model runs in f32 and bf16: every loss,
gradient and BatchNorm buffer must stay finite and equal to the step without the fill."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
import bench  # noqa: E402


def one_step(model, dtype, fill, batch_size=2):
    torch.manual_seed(0)
    torch.use_deterministic_algorithms(fill, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = fill
    dev = torch.device("cuda:0")
    task, opt = bench.build_task(model, dev, False, 0)
    for m in task.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
        if hasattr(m, "dropout_ratio"):
            m.dropout_ratio = 0.0
    batch = bench.synthetic_batch(batch_size, dev, 43)
    task.train()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
        loss = task.training_step(batch, 0)
    loss.backward()
    out = {"loss": loss.detach().float().cpu()}
    for n, p in task.named_parameters():
        if p.grad is not None:
            out["grad:" + n] = p.grad.detach().float().cpu()
    for n, b in task.named_buffers():
        if b.dtype.is_floating_point:
            out["buf:" + n] = b.detach().float().cpu()
    torch.use_deterministic_algorithms(False)
    torch.utils.deterministic.fill_uninitialized_memory = False
    return out


for model in sys.argv[1:] or ["dofa", "segformer", "unetpp"]:
    for dtype in ("f32", "bf16"):
        ref = one_step(model, dtype, False)
        got = one_step(model, dtype, True)
        bad = [k for k, v in got.items() if not torch.isfinite(v).all()]
        diff = [k for k, v in got.items() if k not in bad and not torch.equal(v, ref[k])]
        print(f"{model} {dtype}: {len(got)} tensors, non-finite with NaN-filled torch.empty: {len(bad)} {bad[:6]}, "
              f"different from the unfilled run: {len(diff)} {diff[:6]}", flush=True)
