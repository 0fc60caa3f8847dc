"""UperNet decoder on MI355X (drop-in for the reference's models/decoders/upernet.py)."""

from __future__ import annotations

import torch
from torch import nn

from geo_deep_learning.models.utils import PPM, ConvModule
from gdlhip import nn as gnn
from gdlhip import ops


class UperNetDecoder(nn.Module):
    """PPM + FPN decoder (upernet.py:9-152), scale_modules=False path."""

    def __init__(self, embed_dim: list[int], pool_scales: tuple[int, ...] = (1, 2, 3, 6), channels: int = 256,
                 *, align_corners: bool = True, scale_modules: bool = False) -> None:
        super().__init__()
        if scale_modules:
            msg = "gdlhip UperNetDecoder: scale_modules (ConvTranspose path) is unused by DOFA (dofa.py:68)"
            raise NotImplementedError(msg)
        if align_corners:
            msg = "gdlhip UperNetDecoder: align_corners=True is not on the hot path (dofa.py:66)"
            raise NotImplementedError(msg)
        self.scale_modules = scale_modules
        self.embed_dim = embed_dim
        self.out_channels = channels
        self.channels = channels
        self.align_corners = align_corners
        self.psp_modules = PPM(pool_scales, self.embed_dim[-1], self.channels, align_corners=align_corners)
        self.bottleneck = ConvModule(self.embed_dim[-1] + len(pool_scales) * self.channels, self.channels, 3,
                                     padding=1, inplace=True)
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for embed_dim_ in self.embed_dim[:-1]:
            self.lateral_convs.append(ConvModule(embed_dim_, self.channels, 1, inplace=False))
            self.fpn_convs.append(ConvModule(self.channels, self.channels, 3, padding=1, inplace=False))
        self.fpn_bottleneck = ConvModule(len(self.embed_dim) * self.channels, self.channels, 3, padding=1,
                                         inplace=True)

    def psp_forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        """upernet.py:103-109: cat([x, up(ppm_s(x))...]) -> 3x3 bottleneck; the upsample writes
        straight into the concat buffer."""
        size = (x.shape[1], x.shape[2])
        cat = gnn.concat_upsample([x, *self.psp_modules.forward_nhwc_lowres(x)], size)
        return self.bottleneck.forward_nhwc(cat)

    def forward_nhwc(self, inputs: list[torch.Tensor]) -> torch.Tensor:
        # lateral 1x1 convolutions and the PPM branches are independent: one group (one SyncBatchNorm message per direction)
        lat_items = [dict(x=inputs[i], conv=m.conv, norm=m.norm) for i, m in enumerate(self.lateral_convs)]
        ppm_items = self.psp_modules.items_nhwc(inputs[-1])
        res = gnn.conv_bn_act_group(lat_items + ppm_items)
        laterals = res[:len(lat_items)]
        x = inputs[-1]
        cat = gnn.concat_upsample([x, *res[len(lat_items):]], (x.shape[1], x.shape[2]))      # upernet.py:103-109
        laterals.append(self.bottleneck.forward_nhwc(cat))
        n = len(laterals)
        for i in range(n - 1, 0, -1):  # top-down: lat[i-1] += up(lat[i])
            laterals[i - 1] = gnn.upsample_add(laterals[i - 1], laterals[i])
        fpn_outs = gnn.conv_bn_act_group([dict(x=laterals[i], conv=self.fpn_convs[i].conv, norm=self.fpn_convs[i].norm)
                                          for i in range(n - 1)])
        fpn_outs.append(laterals[-1])
        size = (fpn_outs[0].shape[1], fpn_outs[0].shape[2])
        # 3x3 bottleneck over the concat of the upsampled levels; in training the upsampled levels' gradients are computed at
        # their own resolution (gdlhip.nn.concat_resize_conv_bn_act)
        return gnn.concat_resize_conv_bn_act(fpn_outs, self.fpn_bottleneck.conv, self.fpn_bottleneck.norm,
                                             relu=self.fpn_bottleneck.act is not None)

    def forward(self, inputs: list[torch.Tensor]) -> torch.Tensor:
        cd = gnn.compute_dtype()
        xs = [gnn.to_compute(ops.as_nhwc(x), cd) for x in inputs]
        return ops.as_nchw(self.forward_nhwc(xs))
