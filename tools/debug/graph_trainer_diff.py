import sys
from functools import partial
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "geo-deep-learning_amd")); sys.path.insert(0, str(ROOT / "tests"))
import test_hip_tasks as T
from gdlhip.trainer import MiniTrainer, seed_everything
from oracle import synthetic_batch
batches = [synthetic_batch(4, 3, 112, 5, s) for s in (1, 2, 3, 4)] + [synthetic_batch(2, 3, 112, 5, 5)]
for bt in batches:
    bt["mask"] = (bt["image"][:, :1] * 1.2 + 2).clamp(0, 4).long()
import itertools
for ragged, with_val in itertools.product((True, False), (True, False)):
  out = {}
  print("== ragged last batch:", ragged, " validation:", with_val)
  for mode in (False, "auto"):
    seed_everything(42)
    _, task = T._dofa_task(optimizer=partial(torch.optim.Adam, lr=1e-3))
    for blk in task.model.encoder.blocks:
        blk.drop_prob = 0.0
    task.model.aux_head.dropout_ratio = 0.0
    tr = MiniTrainer(max_epochs=2, precision="32", gradient_clip_val=1.0, default_root_dir="/tmp/gt_" + str(mode), graph_step=mode)
    losses = []
    orig = tr._collect
    def spy(name, value, bs, orig=orig, losses=losses):
        if name == "train_loss":
            losses.append(value.detach().clone())
        orig(name, value, bs)
    tr._collect = spy
    tr.fit(task, train_dataloaders=batches if ragged else batches[:4], val_dataloaders=[batches[0]] if with_val else None)
    out[mode] = [l.item() for l in losses]
    print(mode, "graphed", tr.graphed_steps, ["%.6f" % v for v in out[mode]])
  print("diff", ["%.2e" % abs(a - b) for a, b in zip(out[False], out["auto"])])
