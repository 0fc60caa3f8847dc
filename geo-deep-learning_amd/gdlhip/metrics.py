"""Segmentation metrics on HIP kernels: drop-in for the ``torchmetrics.segmentation.MeanIoU`` /
``torchmetrics.wrappers.ClasswiseWrapper`` pair the reference's tasks build
(tasks_with_models/segmentation_dofa.py:71-85, used in ``test_step`` :313).

The per-sample / per-class intersection and cardinality COUNTS are exact integers from ``gdl_iou_counts`` (LDS
histograms, integer atomics); the few host-side divisions follow torchmetrics 1.8.2
``functional/segmentation/mean_iou.py`` (third-party, not under /root/reference: restated -- "parity unpinned"):
``iou[b,k] = I/U`` with 0 where ``U == 0``; ``update`` adds the batch mean, ``compute`` divides by the number of updates.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from . import ops


class MeanIoU(nn.Module):
    def __init__(self, num_classes: int, include_background: bool = True, per_class: bool = False,
                 input_format: str = "index", **kwargs: object) -> None:
        super().__init__()
        if input_format != "index" or kwargs:
            msg = "gdlhip MeanIoU implements input_format='index' (the reference's configuration)"
            raise NotImplementedError(msg)
        self.num_classes, self.include_background, self.per_class = num_classes, include_background, per_class
        self.register_buffer("score", torch.zeros(num_classes if per_class else 1, dtype=torch.float32),
                             persistent=False)
        self.register_buffer("num_batches", torch.zeros((), dtype=torch.int64), persistent=False)

    def _batch_score(self, preds: Tensor, target: Tensor) -> Tensor:
        counts = ops.iou_counts(preds.long(), target.long(), self.num_classes)     # [B, 3, K] int64
        inter, union = counts[:, 0], counts[:, 1] + counts[:, 2] - counts[:, 0]
        if not self.include_background:
            inter, union = inter[:, 1:], union[:, 1:]
        iou = torch.where(union > 0, inter.float() / union.clamp_min(1).float(), torch.zeros((), device=inter.device))
        return iou.mean(0) if self.per_class else iou.mean().reshape(1)

    def update(self, preds: Tensor, target: Tensor) -> None:
        s = self._batch_score(preds, target)
        if not self.include_background and self.per_class:
            s = torch.cat([torch.zeros(1, device=s.device), s])
        self.score += s.to(self.score.device)
        self.num_batches += 1

    def compute(self) -> Tensor:
        out = self.score / self.num_batches.clamp_min(1)
        return out if self.per_class else out[0]

    def reset(self) -> None:
        self.score.zero_()
        self.num_batches.zero_()

    def forward(self, preds: Tensor, target: Tensor) -> Tensor:
        """torchmetrics ``Metric.forward``: the value for this batch alone, while accumulating the global state."""
        self.update(preds, target)
        s = self._batch_score(preds, target)
        return s if self.per_class else s[0]


class ClasswiseWrapper(nn.Module):
    """``{f"{prefix}{label}": value}`` view of a per-class metric (torchmetrics.wrappers.ClasswiseWrapper)."""

    def __init__(self, metric: MeanIoU, labels: list[str] | None = None, prefix: str | None = None) -> None:
        super().__init__()
        self.metric, self.labels = metric, labels
        self.prefix = prefix if prefix is not None else f"{type(metric).__name__.lower()}_"

    def _named(self, values: Tensor) -> dict[str, Tensor]:
        labels = self.labels or [str(i) for i in range(values.numel())]
        return {f"{self.prefix}{lab}": v for lab, v in zip(labels, values)}

    def forward(self, preds: Tensor, target: Tensor) -> dict[str, Tensor]:
        return self._named(self.metric(preds, target))

    def update(self, preds: Tensor, target: Tensor) -> None:
        self.metric.update(preds, target)

    def compute(self) -> dict[str, Tensor]:
        return self._named(self.metric.compute())

    def reset(self) -> None:
        self.metric.reset()
