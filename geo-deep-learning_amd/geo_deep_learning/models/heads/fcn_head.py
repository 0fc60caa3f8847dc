"""FCN auxiliary head (drop-in for the reference's models/heads/fcn_head.py)."""

from __future__ import annotations

import torch
from torch import nn

from geo_deep_learning.models.utils import ConvModule
from gdlhip import nn as gnn
from gdlhip import ops


class FCNHead(nn.Module):
    """3x3 ConvModule -> Dropout2d -> 1x1 classifier (fcn_head.py:9-84; num_convs=1)."""

    def __init__(self, in_channels: int, channels: int = 256, in_index: int = -1, num_convs: int = 2,
                 num_classes: int = 19, dropout_ratio: float = 0.1, *, concat_input: bool = False) -> None:
        super().__init__()
        if num_convs != 1 or concat_input:
            msg = "gdlhip FCNHead: the DOFA aux head uses num_convs=1, concat_input=False (dofa.py:70-75)"
            raise NotImplementedError(msg)
        self.in_channels = in_channels
        self.channels = channels
        self.in_index = in_index
        self.num_classes = num_classes
        self.concat_input = concat_input
        self.convs = nn.Sequential(ConvModule(in_channels, channels, kernel_size=3, padding=1, inplace=True))
        self.dropout_ratio = dropout_ratio
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else nn.Identity()
        self.cls_seg = nn.Conv2d(channels, num_classes, kernel_size=1)

    def forward_logits(self, x_nhwc: torch.Tensor, size, drop_mask: torch.Tensor | None = None, lowres: bool = False):
        """convs -> Dropout2d (per (sample, channel) scale folded into the classifier) -> 1x1 ->
        bilinear to ``size``.  ``drop_mask`` [B, channels] of 0/1 pins the draw (tests)."""
        feats = self.convs[0].forward_nhwc(x_nhwc)
        chan_scale = None
        if self.training and self.dropout_ratio > 0:
            keep = 1.0 - self.dropout_ratio
            if drop_mask is None:
                drop_mask = torch.empty((feats.shape[0], self.channels), device=feats.device,
                                        dtype=torch.float32).bernoulli_(keep)
            chan_scale = (drop_mask.to(device=feats.device, dtype=torch.float32) / keep).contiguous()
        return gnn.head_logits(feats, self.cls_seg, size, chan_scale, lowres=lowres)

    def forward(self, inputs: torch.Tensor | list[torch.Tensor]) -> torch.Tensor:
        x = inputs[self.in_index] if isinstance(inputs, (list, tuple)) else inputs
        xn = gnn.to_compute(ops.as_nhwc(x), gnn.compute_dtype())
        return self.forward_logits(xn, (xn.shape[1], xn.shape[2]))
