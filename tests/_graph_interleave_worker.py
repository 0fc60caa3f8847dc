"""Worker of tests/test_hip_tasks.py::test_graphed_bf16_step_with_eager_steps_in_between_keeps_derived_operands_current: bf16 training
steps of the tiny DOFA task from a hipGraph with eager steps in between, against an all-eager twin; prints one JSON line.
Runs in its own process (see the test's docstring)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "geo-deep-learning_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402

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


def main():
    torch.manual_seed(1234)
    batches = [T._to_dev(T.synthetic_batch(2, 3, 112, 5, 50 + i)) for i in range(6)]
    for b in batches:
        b["mask"] = b["mask"].long()
    te, oe = make(False)
    tg, og = make(True)
    graphed = GraphedTrainStep(tg, og, batches[0], autocast_dtype=torch.bfloat16, warmup=2)
    cared = 0 if og._repack is None else int(og._repack[1].shape[0])
    for _ in range(2):                                                  # the capture's two warm-up steps were real steps
        eager_step(te, oe, batches[0])
    worst = 0.0
    for i, b in enumerate(batches):
        le = eager_step(te, oe, b).item()
        lg = (eager_step(tg, og, b) if i in (2, 4) else graphed(b)).item()
        worst = max(worst, abs(le - lg) / max(1.0, abs(le)))
    checked = wrong = 0
    for p in tg.parameters():
        for key, val, mode, c0, c1 in gnn.derived_operands(p):
            hit = gnn._CACHE[key]
            if hit[0] != ((p._version, gnn._RAW_WRITES.get(id(p), 0), p.data_ptr()),):
                continue                                                # (stale entries are rebuilt on use)
            m = p.detach().permute(0, 2, 3, 1).reshape(p.shape[0], -1, p.shape[1])
            if mode == gnn.REPACK_SLICE:
                want = m[:, :, c0:c1].reshape(p.shape[0], -1)
            elif mode == gnn.REPACK_TAPS:
                want = m[:, :, c0:c1].permute(1, 0, 2).reshape(-1, c1 - c0)
            else:
                want = p.detach().permute(1, 2, 3, 0).flip(1, 2).reshape(p.shape[1], -1)
            checked += 1
            wrong += int(not torch.equal(val, want.to(torch.bfloat16)))
    print(json.dumps({"operands_under_the_optimizers_care": cared, "max_relative_loss_difference": worst,
                      "derived_operands_checked": checked, "derived_operands_wrong": wrong}))


if __name__ == "__main__":
    main()
