"""SegFormer all-MLP decoder on MI355X (drop-in for the reference's models/decoders/segformer_mlp.py)."""

from __future__ import annotations

import torch
from torch import nn

from geo_deep_learning.models.utils import _cl_conv
from gdlhip import nn as gnn
from gdlhip import ops, tnn


class MLP(nn.Module):
    """Linear embedding of one pyramid level (segformer_mlp.py:8-19)."""

    def __init__(self, input_dim: int = 2048, embed_dim: int = 768) -> None:
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        """[B,h,w,C] -> [B,h,w,E] (the reference's flatten/transpose is a no-op in NHWC)."""
        return tnn.conv(x, self.proj.weight, self.proj.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xn = gnn.to_compute(ops.as_nhwc(x), gnn.compute_dtype())
        y = self.forward_nhwc(xn)
        return y.reshape(y.shape[0], -1, y.shape[-1])


class Decoder(nn.Module):
    """segformer_mlp.py:22-130: 4 x Linear -> bilinear to 1/4 res -> concat -> 1x1 conv + BN + ReLU ->
    Dropout2d -> 1x1 classifier.  The 1x1 fusion convolution commutes with the resizes: it runs per level at the level's
    own resolution and the upsampled partial results are summed (gdlhip.nn.pyramid_fuse_bn_act) -- no concat buffer."""

    def __init__(self, encoder: str = "mit_b2", in_channels: list[int] | None = None,
                 feature_strides: list[int] | None = None, embedding_dim: int = 768, num_classes: int = 1,
                 dropout_ratio: float = 0.1) -> None:
        super().__init__()
        if feature_strides is None:
            feature_strides = [4, 8, 16, 32]
        if in_channels is None:
            in_channels = [64, 128, 320, 512]
        if encoder == "mit_b0":
            in_channels = [32, 64, 160, 256]
            embedding_dim = 256
        elif encoder == "mit_b1":
            embedding_dim = 256
        if len(feature_strides) != len(in_channels):
            msg = "feature_strides and in_channels must have the same length"
            raise ValueError(msg)
        if min(feature_strides) != feature_strides[0]:
            msg = "The minimum feature stride must be the first element"
            raise ValueError(msg)
        self.num_classes = num_classes
        self.in_channels = in_channels
        c1, c2, c3, c4 = in_channels
        self.linear_c4 = MLP(input_dim=c4, embed_dim=embedding_dim)
        self.linear_c3 = MLP(input_dim=c3, embed_dim=embedding_dim)
        self.linear_c2 = MLP(input_dim=c2, embed_dim=embedding_dim)
        self.linear_c1 = MLP(input_dim=c1, embed_dim=embedding_dim)
        self.linear_fuse = nn.Sequential(_cl_conv(embedding_dim * 4, embedding_dim, 1, padding=0, bias=False),
                                         nn.BatchNorm2d(embedding_dim), nn.ReLU(inplace=True))
        self.dropout_ratio = dropout_ratio
        self.dropout = nn.Dropout2d(dropout_ratio)
        self.linear_pred = nn.Conv2d(embedding_dim, self.num_classes, kernel_size=1)

    def forward_logits(self, feats: list[torch.Tensor], size, drop_mask: torch.Tensor | None = None) -> torch.Tensor:
        """NHWC stage features -> NCHW f32 logits at ``size`` (decoder + the model's final resize fused)."""
        c1, c2, c3, c4 = feats
        lv = [self.linear_c4.forward_nhwc(c4), self.linear_c3.forward_nhwc(c3), self.linear_c2.forward_nhwc(c2),
              self.linear_c1.forward_nhwc(c1)]
        # linear_fuse over the concat of the upsampled levels, evaluated per level at the level's own resolution
        fused = gnn.pyramid_fuse_bn_act(lv, self.linear_fuse[0], self.linear_fuse[1], relu=True)
        chan_scale = None
        if self.training and self.dropout_ratio > 0:
            keep = 1.0 - self.dropout_ratio
            if drop_mask is None:
                drop_mask = torch.empty((fused.shape[0], fused.shape[-1]), device=fused.device,
                                        dtype=torch.float32).bernoulli_(keep)
            chan_scale = (drop_mask.to(device=fused.device, dtype=torch.float32) / keep).contiguous()
        return gnn.head_logits(fused, self.linear_pred, size, chan_scale)

    def forward(self, x: list[torch.Tensor]) -> torch.Tensor:
        cd = gnn.compute_dtype()
        feats = [gnn.to_compute(ops.as_nhwc(f), cd) for f in x]
        return self.forward_logits(feats, (feats[0].shape[1], feats[0].shape[2]))
