"""hipGraph capture of whole steps (static shapes): one graph replay instead of several hundred kernel launches issued
one by one through ctypes.

At the reference's own per-GPU batch of 4 (configs/dofa_config_RGB.yaml:85) a DOFA-base + UperNet training step is ~600
launches of 5-20 us each: the GPU finishes them faster than the host can issue them (batch 2 and batch 4 took the same
9.7 ms).  A captured step has no host work per kernel.  Everything the step does runs on torch's current stream through
the C-ABI, allocates through torch's caching allocator (graph-private pool under capture) and keeps per-step scalars on
the device (``FusedAdam(capturable=True)``: step count, bias corrections, learning rate), so ``torch.cuda.graph`` can
record forward + loss + backward + clipping + Adam as they are.

Rules (torch.cuda.graphs): static input buffers (new batches are COPIED into them), no host read-backs inside the step, a
few eager warm-up steps on a side stream first (lazy initialisations, cached weight casts, kernel attributes).  DropPath /
Dropout2d draws use torch's graph-safe Philox generator: each replay draws fresh masks.
"""

from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from . import nn as gnn


def _clone_static(batch: dict[str, Any]) -> dict[str, Any]:
    return {k: (v.clone() if isinstance(v, Tensor) and v.is_cuda else v) for k, v in batch.items()}


def _copy_into(static: dict[str, Any], batch: dict[str, Any]) -> None:
    for k, v in batch.items():
        s = static.get(k)
        if isinstance(s, Tensor) and s.is_cuda:
            if not isinstance(v, Tensor) or v.shape != s.shape or v.dtype != s.dtype:
                msg = f"graphed step: batch[{k!r}] must keep shape {tuple(s.shape)} / dtype {s.dtype} of the captured batch"
                raise ValueError(msg)
            s.copy_(v, non_blocking=True)


class GraphedTrainStep:
    """``loss = step(batch)`` == zero_grad -> autocast(training_step) -> backward -> optimizer.step(), replayed from a hipGraph.

    ``task``: a LightningModule-shaped task (``training_step(batch, idx) -> loss``); ``optimizer``: ``FusedAdam`` created with
    ``capturable=True``.  The returned loss is a static device tensor (read it after the next synchronisation point)."""

    def __init__(self, task, optimizer, example_batch: dict[str, Any], *, autocast_dtype: torch.dtype | None = torch.bfloat16,
                 warmup: int = 3) -> None:
        if not getattr(optimizer, "capturable", False):
            msg = "GraphedTrainStep needs FusedAdam(capturable=True): step count and learning rate must live on the device"
            raise ValueError(msg)
        self.task, self.optimizer, self.autocast_dtype = task, optimizer, autocast_dtype
        self.static = _clone_static(example_batch)
        task.train()
        # gradients (and their AccumulateGrad nodes) of earlier eager steps belong to another stream: start clean
        optimizer.zero_grad(set_to_none=True)
        if isinstance(getattr(task, "logged", None), dict):
            task.logged.clear()
        # a trainer's metric sink accumulates logged tensors on the device: captured once, it would add the SAME static buffer at
        # every replay while the host-side weights stood still.  Logging is detached for warm-up and capture; after a replay the
        # caller logs `step.loss` itself (MiniTrainer does)
        trainer = getattr(task, "trainer", None)
        sink = getattr(trainer, "_collect", None)
        if sink is not None:
            trainer._collect = lambda *a, **k: None
        try:
            self._capture(task, optimizer, warmup)
        finally:
            if sink is not None:
                trainer._collect = sink

    def _capture(self, task, optimizer, warmup: int) -> None:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)           # gradients are (re)allocated from the graph's private pool
        with torch.cuda.graph(self.graph):
            self.loss = self._eager(zero=False)
        self._rewritten = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self._rewritten += [b for m in task.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training
                            for b in (m.running_mean, m.running_var) if b is not None]

    def _eager(self, zero: bool = True) -> Tensor:
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            loss = self.task.training_step(self.static, 0)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, batch: dict[str, Any] | None = None) -> Tensor:
        """Returns the STATIC loss buffer: the next replay overwrites it (clone it to keep a value)."""
        if batch is not None and batch is not self.static:
            _copy_into(self.static, batch)
        self.optimizer.sync_lr()        # per-step schedulers (OneCycleLR) write param_groups["lr"] on the host
        self.graph.replay()
        self.optimizer.note_replay()
        for t in self._rewritten:       # the replay rewrote these through raw pointers: eager code must not trust operands
            gnn.mark_updated(t)         # cached from their earlier values (eval-time BatchNorm folds, packed weights)
        return self.loss


class GraphedEvalStep:
    """``out = step(batch)`` == no_grad + autocast(fn(batch)) replayed from a hipGraph; ``fn`` e.g. ``lambda b: task(b["image"],
    b["wavelengths"])`` or a validation step that returns tensors (their storage is static: copy what must outlive a replay)."""

    def __init__(self, fn, example_batch: dict[str, Any], *, autocast_dtype: torch.dtype | None = torch.bfloat16, warmup: int = 2) -> None:
        self.fn, self.autocast_dtype = fn, autocast_dtype
        self.static = _clone_static(example_batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._eager()

    def _eager(self):
        with torch.no_grad(), torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            return self.fn(self.static)

    def __call__(self, batch: dict[str, Any] | None = None):
        if batch is not None and batch is not self.static:
            _copy_into(self.static, batch)
        self.graph.replay()
        return self.out
