#!/usr/bin/env python
"""What happens to the process when a hipGraph capture of the training step raises: (a) before anything was recorded (an
empty capture), (b) after the forward was recorded.  Prints progress line by line (a crash shows where).
    python tools/debug/r05_failed_capture.py empty|midway|item"""
import faulthandler
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "geo-deep-learning_amd"), str(ROOT / "tests")]
faulthandler.enable()
mode = sys.argv[1] if len(sys.argv) > 1 else "empty"


def say(*a):
    print(*a, flush=True)


import test_hip_tasks as T  # noqa: E402
from gdlhip import nn as gnn  # noqa: E402
from gdlhip.graphs import GraphCaptureFatal, GraphedTrainStep  # noqa: E402

_, task = T._dofa_task(freeze=("encoder",))
task.trainer = T._Trainer(True)
params = [p for p in task.parameters() if p.requires_grad]
opt = gnn.FusedAdam(params, lr=1e-2, max_grad_norm=1.0, capturable=True)
batch = T._to_dev(T.synthetic_batch(2, 3, 112, 5, 77))
batch["mask"] = batch["mask"].long()
task.train()
task.training_step(batch, 0).backward()
opt.step()
opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
say("eager step done")
real = task.training_step


def failing(b, i):
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and mode == "empty":
        raise RuntimeError("cannot record (nothing captured yet)")
    loss = real(b, i)
    if capturing and mode == "midway":
        raise RuntimeError("cannot record (after the forward)")
    if capturing and mode == "item":
        loss.item()          # a host read-back under capture: HIP error
    return loss


task.training_step = failing
try:
    GraphedTrainStep(task, opt, batch, autocast_dtype=None, warmup=2, restore_state=True)
    say("capture unexpectedly succeeded")
except GraphCaptureFatal as exc:
    say("capture raised the FATAL error (expected for mode item):", str(exc)[:300])
    say("parameters restored? device step", float(opt.device_state(0)[0]))
    sys.exit(0)
except BaseException as exc:  # noqa: BLE001
    say("capture raised:", type(exc).__name__, str(exc)[:200])
task.training_step = real
say("capturing now?", torch.cuda.is_current_stream_capturing())
torch.cuda.synchronize()
say("synchronised; device step", float(opt.device_state(0)[0]))
loss = task.training_step(batch, 0)
loss.backward()
opt.step()
opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
say("eager step after the failure ok, loss", float(loss.detach()))
del loss      # (a live autograd graph keeps default-stream AccumulateGrad nodes: a later capture then crashes in hipStreamEndCapture)
gs = GraphedTrainStep(task, opt, batch, autocast_dtype=None, warmup=2, restore_state=True)
for _ in range(3):
    gs(batch)
torch.cuda.synchronize()
say("second capture + 3 replays ok; samples", task.train_samples_count, "device step", float(opt.device_state(0)[0]))
