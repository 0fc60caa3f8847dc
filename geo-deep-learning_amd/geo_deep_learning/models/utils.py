"""ConvModule / PPM / resize on MI355X (drop-in for the reference's models/utils.py).

Internally every tensor is NHWC in the compute dtype; ``ConvModule.forward`` keeps the
reference's NCHW-in / NCHW-out contract by taking and returning channels-last views, while the
fused ``forward_nhwc`` is what the neck / decoder call.
"""

from __future__ import annotations

import torch
from torch import nn

from gdlhip import nn as gnn
from gdlhip import ops


def _cl_conv(cin: int, cout: int, k: int, *, padding: int, bias: bool) -> nn.Conv2d:
    """nn.Conv2d parameter container whose weight is STORED channels-last ([N,R,S,C] in memory ==
    the implicit-GEMM operand), logical shape / state-dict layout unchanged."""
    conv = nn.Conv2d(cin, cout, k, padding=padding, bias=bias)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


class ConvModule(nn.Module):
    """conv(bias=False) -> BatchNorm2d -> ReLU (models/utils.py:10-52)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, padding: int = 0,
                 dilation: int = 1, stride: int = 1, *, inplace: bool = False, transpose: bool = False,
                 scale_factor: int | None = None) -> None:
        super().__init__()
        if transpose or dilation != 1 or stride != 1:
            msg = "gdlhip ConvModule: only stride-1, dilation-1, non-transposed convs are on the hot path"
            raise NotImplementedError(msg)
        self.conv = _cl_conv(in_channels, out_channels, kernel_size, padding=padding, bias=False)
        self.norm = nn.BatchNorm2d(out_channels)
        self.act = nn.ReLU(inplace=inplace)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        return gnn.conv_bn_act(x, self.conv, self.norm, relu=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = gnn.to_compute(ops.as_nhwc(x), gnn.compute_dtype())
        return ops.as_nchw(self.forward_nhwc(x))


class PPM(nn.ModuleList):
    """Pooling Pyramid Module (models/utils.py:55-93)."""

    def __init__(self, pool_scales: tuple[int, ...], in_channels: int, channels: int, *,
                 align_corners: bool) -> None:
        super().__init__()
        if align_corners:
            msg = "gdlhip PPM: align_corners=True is not on the hot path (dofa.py:66 uses False)"
            raise NotImplementedError(msg)
        self.pool_scales = pool_scales
        self.align_corners = align_corners
        self.in_channels = in_channels
        self.channels = channels
        for pool_scale in pool_scales:
            self.append(nn.Sequential(nn.AdaptiveAvgPool2d(pool_scale),
                                      ConvModule(self.in_channels, self.channels, 1, inplace=True)))

    def items_nhwc(self, x: torch.Tensor) -> list[dict]:
        """Per scale: the pooled map and the ConvModule that follows it, as members for gnn.conv_bn_act_group (the branches
        are independent of each other and of UperNet's lateral convolutions)."""
        items = []
        for ppm in self:
            pooled = gnn.adaptive_avgpool(x, int(ppm[0].output_size if isinstance(ppm[0].output_size, int)
                                                 else ppm[0].output_size[0]))
            items.append(dict(x=pooled, conv=ppm[1].conv, norm=ppm[1].norm))
        return items

    def forward_nhwc_lowres(self, x: torch.Tensor) -> list[torch.Tensor]:
        """Per-scale [B,s,s,channels] outputs BEFORE the upsample (the caller fuses upsample+concat)."""
        return gnn.conv_bn_act_group(self.items_nhwc(x))

    def forward(self, x: torch.Tensor) -> list[torch.Tensor]:
        xn = gnn.to_compute(ops.as_nhwc(x), gnn.compute_dtype())
        size = (xn.shape[1], xn.shape[2])
        return [ops.as_nchw(gnn.bilinear(o, size)) for o in self.forward_nhwc_lowres(xn)]


def resize_nhwc(x: torch.Tensor, scale_factor: float) -> torch.Tensor:
    """``resize(scale_factor=s, mode='bilinear')`` (models/utils.py:96-137): size = int(h*s)."""
    h, w = x.shape[1], x.shape[2]
    return gnn.bilinear(x, (int(h * scale_factor), int(w * scale_factor)))


def resize(input_: torch.Tensor, size=None, scale_factor=None, mode: str = "nearest", *,
           align_corners: bool | None = None, warning: bool = True) -> torch.Tensor:
    """NCHW wrapper with the reference signature; only the bilinear/align_corners=False path."""
    if mode != "bilinear" or align_corners:
        msg = "gdlhip resize: only mode='bilinear', align_corners in (None, False)"
        raise NotImplementedError(msg)
    x = gnn.to_compute(ops.as_nhwc(input_), gnn.compute_dtype())
    if scale_factor is not None:
        size = (int(x.shape[1] * scale_factor), int(x.shape[2] * scale_factor))
    return ops.as_nchw(gnn.bilinear(x, size))
