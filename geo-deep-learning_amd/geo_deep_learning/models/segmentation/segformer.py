"""SegFormer segmentation model on MI355X (drop-in for the reference's models/segmentation/segformer.py)."""

from __future__ import annotations

import torch

from geo_deep_learning.models.decoders.segformer_mlp import Decoder
from geo_deep_learning.models.encoders.mix_transformer import DynamicMixTransformer, get_encoder
from gdlhip import nn as gnn

from .base import BaseSegmentationModel


class SegFormerSegmentationModel(BaseSegmentationModel):
    """MiT encoder -> all-MLP decoder -> bilinear x4 (segformer.py:15-57)."""

    def __init__(self, encoder: str = "mit_b0", in_channels: int = 3, weights: str | None = None,
                 freeze_layers: list[str] | None = None, num_classes: int = 1, *,
                 use_dynamic_encoder: bool = False) -> None:
        super().__init__()
        if use_dynamic_encoder:
            self.encoder = DynamicMixTransformer(encoder=encoder, weights=weights)
        else:
            self.encoder = get_encoder(name=encoder, in_channels=in_channels, depth=5, weights=weights)
        if freeze_layers:
            self._freeze_layers(layers=freeze_layers)
        self.decoder = Decoder(encoder=encoder, num_classes=num_classes)

    def forward(self, img: torch.Tensor, drop_masks=None, dec_drop_mask: torch.Tensor | None = None) -> torch.Tensor:
        with gnn.counter_batch():       # the BatchNorm step counters of the pass advance in one launch
            feats = self.encoder.forward_nhwc(img, drop_masks)
            return self.decoder.forward_logits(feats, img.shape[2:], dec_drop_mask)
