"""MultiLevelNeck on MI355X (drop-in for the reference's models/necks/multilevel_neck.py)."""

from __future__ import annotations

import torch
from torch import nn

from geo_deep_learning.models.utils import _cl_conv, resize_nhwc
from gdlhip import nn as gnn
from gdlhip import ops


class ConvModule(nn.Module):
    """conv (WITH bias) -> BN -> ReLU (multilevel_neck.py:28-67)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int,
                 norm_cfg: dict | None = None, act_cfg: dict | None = None, stride: int = 1,
                 padding: int = 0) -> None:
        super().__init__()
        if stride != 1:
            msg = "gdlhip neck ConvModule: stride must be 1"
            raise NotImplementedError(msg)
        self.conv = _cl_conv(in_channels, out_channels, kernel_size, padding=padding, bias=True)
        self.norm = nn.BatchNorm2d(out_channels) if norm_cfg is not None and norm_cfg.get("type") == "BN" else None
        self.act = nn.ReLU(inplace=True) if act_cfg is not None and act_cfg.get("type") == "ReLU" else None
        if self.norm is None:
            msg = "gdlhip neck ConvModule: norm_cfg={'type': 'BN'} is required (dofa.py:58-64)"
            raise NotImplementedError(msg)

    def forward_nhwc(self, x: torch.Tensor, *, up4: bool = False, up: int = 1) -> torch.Tensor:
        """``up`` (2 / 4; ``up4`` = 4): x is first upsampled by that factor (bilinear, align_corners=False) inside the node."""
        return gnn.conv_bn_act(x, self.conv, self.norm, relu=self.act is not None, up4=up4, up=up)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = gnn.to_compute(ops.as_nhwc(x), gnn.compute_dtype())
        return ops.as_nchw(self.forward_nhwc(x))


class MultiLevelNeck(nn.Module):
    """ViT -> feature pyramid neck (multilevel_neck.py:70-160)."""

    def __init__(self, in_channels: list[int], out_channels: list[int], scales: list[float] | None = None,
                 norm_cfg: dict | None = None, act_cfg: dict | None = None) -> None:
        super().__init__()
        if not isinstance(in_channels, list):
            msg = f"in_channels must be a list, but got {type(in_channels)}"
            raise TypeError(msg)
        if not isinstance(out_channels, list):
            msg = f"out_channels must be a list, but got {type(out_channels)}"
            raise TypeError(msg)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.scales = scales or [0.5, 1, 2, 4]
        self.num_outs = len(self.scales)
        self.lateral_convs = nn.ModuleList(
            [ConvModule(ci, co, kernel_size=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
             for ci, co in zip(in_channels, out_channels)])
        self.convs = nn.ModuleList(
            [ConvModule(co, co, kernel_size=3, padding=1, stride=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
             for co in out_channels])

    def forward_nhwc(self, inputs: list[torch.Tensor]) -> list[torch.Tensor]:
        # the lateral convolutions are independent of each other, and so are the 3x3 convolutions: each set runs as one group,
        # so that under SyncBatchNorm their statistics travel in ONE message per direction (gnn.conv_bn_act_group)
        lat = gnn.conv_bn_act_group([dict(x=inputs[i], conv=m.conv, norm=m.norm, relu=m.act is not None)
                                     for i, m in enumerate(self.lateral_convs)])
        if len(lat) == 1:
            lat = [lat[0] for _ in range(self.num_outs)]
        items = []
        for i in range(self.num_outs):
            m = self.convs[i]
            if self.scales[i] in (2, 4):
                # resize x4 / x2 -> 3x3 conv as ONE fused op: forward through the nine low-resolution tap products (no
                # [B,4H,4W,C] intermediate, 1/16 resp. 1/4 of the MACs), both gradients at the low resolution
                items.append(dict(x=lat[i], conv=m.conv, norm=m.norm, relu=m.act is not None, up=int(self.scales[i])))
            else:
                items.append(dict(x=resize_nhwc(lat[i], self.scales[i]), conv=m.conv, norm=m.norm, relu=m.act is not None))
        return gnn.conv_bn_act_group(items)

    def forward(self, inputs: list[torch.Tensor]) -> tuple[torch.Tensor, ...]:
        if len(inputs) != len(self.in_channels):
            msg = (f"len(inputs) must be equal to len(in_channels), "
                   f"but got {len(inputs)} and {len(self.in_channels)}")
            raise ValueError(msg)
        cd = gnn.compute_dtype()
        xs = [gnn.to_compute(ops.as_nhwc(x), cd) for x in inputs]
        return tuple(ops.as_nchw(o) for o in self.forward_nhwc(xs))
