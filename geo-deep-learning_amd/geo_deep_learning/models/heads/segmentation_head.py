"""Segmentation head (drop-in for the reference's models/heads/segmentation_head.py)."""

from typing import NamedTuple

import torch
from torch import nn

from gdlhip import nn as gnn
from gdlhip import ops


class SegmentationOutput(NamedTuple):
    """Segmentation output (segmentation_head.py:9-13)."""

    out: torch.Tensor
    aux: torch.Tensor | None


class SegmentationHead(nn.Module):
    """1x1 convolution classifier (segmentation_head.py:16-26)."""

    def __init__(self, in_channels: int, num_classes: int) -> None:
        super().__init__()
        self.conv = nn.Conv2d(in_channels, num_classes, kernel_size=1)

    def forward_logits(self, x_nhwc: torch.Tensor, size, lowres: bool = False):
        """head conv + bilinear resize to ``size`` fused: NCHW f32 logits (dofa.py:89-96); ``lowres``: gdlhip.nn.LowresLogits."""
        return gnn.head_logits(x_nhwc, self.conv, size, lowres=lowres)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xn = gnn.to_compute(ops.as_nhwc(x), gnn.compute_dtype())
        return self.forward_logits(xn, (xn.shape[1], xn.shape[2]))
