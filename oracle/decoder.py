"""Oracle restatement of MultiLevelNeck / PPM / UperNetDecoder / heads.

TEST INFRASTRUCTURE ONLY.  Follows, with identical state-dict keys:
  /root/reference/geo_deep_learning/models/necks/multilevel_neck.py
  /root/reference/geo_deep_learning/models/utils.py
  /root/reference/geo_deep_learning/models/decoders/upernet.py
  /root/reference/geo_deep_learning/models/heads/{fcn_head,segmentation_head}.py
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class NeckConvModule(nn.Module):
    """conv (WITH bias) -> BN -> ReLU.  Reference: multilevel_neck.py:28-67."""

    def __init__(self, cin: int, cout: int, k: int, padding: int = 0) -> None:
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=1, padding=padding)
        self.norm = nn.BatchNorm2d(cout)

    def forward(self, x: Tensor) -> Tensor:
        return F.relu(self.norm(self.conv(x)))


class ConvModule(nn.Module):
    """conv (bias=False) -> BN -> ReLU.  Reference: models/utils.py:10-52."""

    def __init__(self, cin: int, cout: int, k: int = 3, padding: int = 0) -> None:
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding, bias=False)
        self.norm = nn.BatchNorm2d(cout)

    def forward(self, x: Tensor) -> Tensor:
        return F.relu(self.norm(self.conv(x)))


class MultiLevelNeck(nn.Module):
    """Reference: multilevel_neck.py:70-160 (scales [4,2,1,0.5], BN+ReLU)."""

    def __init__(self, in_channels: list[int], out_channels: list[int],
                 scales: list[float]) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.scales = scales
        self.lateral_convs = nn.ModuleList(
            [NeckConvModule(ci, co, 1) for ci, co in zip(in_channels, out_channels)])
        self.convs = nn.ModuleList(
            [NeckConvModule(co, co, 3, padding=1) for co in out_channels])

    def forward(self, inputs: list[Tensor]) -> tuple[Tensor, ...]:
        if len(inputs) != len(self.in_channels):
            msg = "len(inputs) must be equal to len(in_channels)"
            raise ValueError(msg)
        lat = [conv(inputs[i]) for i, conv in enumerate(self.lateral_convs)]
        outs = []
        for i, s in enumerate(self.scales):
            h, w = lat[i].shape[2:]
            # models/utils.py:106-137: size=(int(h*s), int(w*s)), align_corners=None
            x = F.interpolate(lat[i], size=(int(h * s), int(w * s)), mode="bilinear",
                              align_corners=None)
            outs.append(self.convs[i](x))
        return tuple(outs)


class PPM(nn.ModuleList):
    """Reference: models/utils.py:55-93."""

    def __init__(self, pool_scales: tuple[int, ...], in_channels: int, channels: int,
                 *, align_corners: bool) -> None:
        super().__init__()
        self.align_corners = align_corners
        for s in pool_scales:
            self.append(nn.Sequential(nn.AdaptiveAvgPool2d(s),
                                      ConvModule(in_channels, channels, 1)))

    def forward(self, x: Tensor) -> list[Tensor]:
        return [F.interpolate(ppm(x), size=x.shape[2:], mode="bilinear",
                              align_corners=self.align_corners) for ppm in self]


class UperNetDecoder(nn.Module):
    """Reference: upernet.py:9-152 (scale_modules=False path)."""

    def __init__(self, embed_dim: list[int], pool_scales=(1, 2, 3, 6), channels: int = 256,
                 *, align_corners: bool = False) -> None:
        super().__init__()
        self.align_corners = align_corners
        self.psp_modules = PPM(pool_scales, embed_dim[-1], channels, align_corners=align_corners)
        self.bottleneck = ConvModule(embed_dim[-1] + len(pool_scales) * channels, channels, 3,
                                     padding=1)
        self.lateral_convs = nn.ModuleList([ConvModule(e, channels, 1) for e in embed_dim[:-1]])
        self.fpn_convs = nn.ModuleList(
            [ConvModule(channels, channels, 3, padding=1) for _ in embed_dim[:-1]])
        self.fpn_bottleneck = ConvModule(len(embed_dim) * channels, channels, 3, padding=1)

    def psp_forward(self, inputs) -> Tensor:
        x = inputs[-1]
        return self.bottleneck(torch.cat([x, *self.psp_modules(x)], dim=1))

    def forward(self, inputs) -> Tensor:
        lat = [conv(inputs[i]) for i, conv in enumerate(self.lateral_convs)]
        lat.append(self.psp_forward(inputs))
        n = len(lat)
        for i in range(n - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(
                lat[i], size=lat[i - 1].shape[2:], mode="bilinear",
                align_corners=self.align_corners)
        outs = [self.fpn_convs[i](lat[i]) for i in range(n - 1)]
        outs.append(lat[-1])
        for i in range(n - 1, 0, -1):
            outs[i] = F.interpolate(outs[i], size=outs[0].shape[2:], mode="bilinear",
                                    align_corners=self.align_corners)
        return self.fpn_bottleneck(torch.cat(outs, dim=1))


class FCNHead(nn.Module):
    """num_convs=1, concat_input=False.  Reference: fcn_head.py:9-84."""

    def __init__(self, in_channels: int, channels: int = 256, num_classes: int = 19,
                 dropout_ratio: float = 0.1) -> None:
        super().__init__()
        self.convs = nn.Sequential(ConvModule(in_channels, channels, 3, padding=1))
        self.dropout_ratio = dropout_ratio
        self.cls_seg = nn.Conv2d(channels, num_classes, kernel_size=1)

    def forward(self, x: Tensor, drop_mask: Tensor | None = None) -> Tensor:
        """``drop_mask`` [B, channels] of 0/1 pins Dropout2d for parity tests."""
        f = self.convs(x)
        if self.training and self.dropout_ratio > 0:
            if drop_mask is None:
                drop_mask = f.new_empty(f.shape[:2]).bernoulli_(1 - self.dropout_ratio)
            f = f * (drop_mask.to(f.dtype) / (1 - self.dropout_ratio))[:, :, None, None]
        return self.cls_seg(f)


class SegmentationHead(nn.Module):
    """1x1 conv classifier.  Reference: segmentation_head.py:16-26."""

    def __init__(self, in_channels: int, num_classes: int) -> None:
        super().__init__()
        self.conv = nn.Conv2d(in_channels, num_classes, kernel_size=1)

    def forward(self, x: Tensor) -> Tensor:
        return self.conv(x)
