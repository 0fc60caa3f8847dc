#!/usr/bin/env python
"""Per-step training losses of MiniTrainer under a one-rank RCCL DDP wrapper: eager vs captured step (where do they part?)."""
import os
import sys
import tempfile
from functools import partial
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "geo-deep-learning_amd"), str(ROOT / "tests")]
import test_hip_tasks as T  # noqa: E402
from gdlhip.trainer import MiniTrainer, seed_everything  # noqa: E402


class Recording(MiniTrainer):
    """Records per logged value: the value, a checksum of the trainable parameters and of the BatchNorm buffers at that moment
    (AFTER the step's update for train_loss), and a checksum of the image the step saw."""

    def _collect(self, name, value, batch_size=None):
        task = self.task_ref
        with torch.no_grad():
            psum = float(sum(p.double().abs().sum() for p in task.parameters() if p.requires_grad))
            bsum = float(sum(b.double().abs().sum() for n, b in task.named_buffers() if "running" in n))
        g = getattr(self, "_graphed", None)
        img = (g.static["image"] if g is not None and name == "train_loss" and self.last_was_replay else self.last_image)
        self.trace.append((name, float(value), psum, bsum, float(img.double().abs().sum()) if img is not None else 0.0))
        return super()._collect(name, value, batch_size)

    def _graph_step(self, model, step_opt, batch, device):
        self.last_image = batch["image"]
        ok = super()._graph_step(model, step_opt, batch, device)
        self.last_was_replay = ok
        return ok


batches = [T.synthetic_batch(4, 3, 112, 5, s) for s in (1, 2, 3, 4)] + [T.synthetic_batch(2, 3, 112, 5, 5)]
for bt in batches:
    bt["mask"] = (bt["image"][:, :1] * 1.2 + 2).clamp(0, 4).long()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29791")
ddp = len(sys.argv) < 2 or sys.argv[1] != "noddp"
if ddp:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
traces = {}
for mode in (False, "auto"):
    seed_everything(42)
    _, task = T._dofa_task(optimizer=partial(torch.optim.Adam, lr=1e-3),
                           scheduler=partial(torch.optim.lr_scheduler.StepLR, step_size=3, gamma=0.5),
                           scheduler_config={"interval": "step", "frequency": 1})
    for blk in task.model.encoder.blocks:
        blk.drop_prob = 0.0
    task.model.aux_head.dropout_ratio = 0.0
    tr = Recording(max_epochs=2, precision="32", gradient_clip_val=1.0, default_root_dir=tempfile.mkdtemp(), graph_step=mode,
                   sync_batchnorm=ddp, force_ddp=ddp)
    tr.trace, tr.task_ref, tr.last_image, tr.last_was_replay = [], task, None, False
    orig_step = task.training_step

    def spy(b, i, _tr=tr, _orig=orig_step):
        _tr.last_image, _tr.last_was_replay = b["image"], False
        return _orig(b, i)
    task.training_step = spy
    tr.fit(task, train_dataloaders=batches, val_dataloaders=[batches[0]])
    traces[mode] = tr.trace
    print(mode, "graphed steps", tr.graphed_steps, flush=True)
for (n0, a, pa, ba, ia), (n1, b, pb, bb, ib) in zip(traces[False], traces["auto"]):
    print(f"{n0:12s} eager {a:.7f}  graphed {b:.7f}  diff {abs(a - b):.2e} | params {pa:.6f} / {pb:.6f} | bn buffers {ba:.6f} / {bb:.6f} "
          f"| image {ia:.4f} / {ib:.4f}" + ("   <--" if abs(a - b) > 1e-6 else ""), flush=True)
if ddp:
    dist.destroy_process_group()
