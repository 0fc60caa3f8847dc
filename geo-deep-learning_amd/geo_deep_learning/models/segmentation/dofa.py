"""DOFA segmentation model on MI355X (drop-in for the reference's models/segmentation/dofa.py)."""

from __future__ import annotations

import torch

from geo_deep_learning.models.decoders.upernet import UperNetDecoder
from geo_deep_learning.models.encoders.dofa_v2 import DOFAv2, create_dofa_base, create_dofa_large
from geo_deep_learning.models.heads.fcn_head import FCNHead
from geo_deep_learning.models.heads.segmentation_head import SegmentationHead, SegmentationOutput
from geo_deep_learning.models.necks.multilevel_neck import MultiLevelNeck
from gdlhip import nn as gnn
from gdlhip import ops
from gdlhip.markers import rng

from .base import BaseSegmentationModel


class DOFASegmentationModel(BaseSegmentationModel):
    """DOFA-ViT encoder -> MultiLevelNeck -> UperNet -> head (+ FCN aux head) (dofa.py:24-107)."""

    def __init__(self, encoder: str = "dofa_base", image_size: tuple[int, int] = (512, 512),
                 freeze_layers: list[str] | None = None, num_classes: int = 1, *,
                 pretrained: bool = True) -> None:
        super().__init__(DOFAv2, MultiLevelNeck, UperNetDecoder, SegmentationHead, SegmentationOutput)
        if encoder == "dofa_base":
            self.embed_dim = 768
            self.encoder = create_dofa_base(img_size=image_size, pretrained=pretrained)
        elif encoder == "dofa_large":
            self.embed_dim = 1024
            self.encoder = create_dofa_large(img_size=image_size, pretrained=pretrained)
        elif isinstance(encoder, DOFAv2):  # a pre-built encoder (small-shape tests)
            self.embed_dim = encoder.embed_dim
            self.encoder = encoder
        else:
            msg = f"Invalid encoder: {encoder}"
            raise ValueError(msg)
        self.neck = MultiLevelNeck(in_channels=[self.embed_dim] * 4, out_channels=[self.embed_dim] * 4,
                                   scales=[4, 2, 1, 0.5], norm_cfg={"type": "BN"}, act_cfg={"type": "ReLU"})
        self.decoder = UperNetDecoder(embed_dim=[self.embed_dim] * 4, pool_scales=(1, 2, 3, 6), channels=256,
                                      align_corners=False, scale_modules=False)
        self.aux_head = FCNHead(in_channels=self.embed_dim, channels=256, num_convs=1, num_classes=num_classes)
        self.head = SegmentationHead(in_channels=256, num_classes=num_classes)
        self.output_struct = SegmentationOutput
        if freeze_layers:
            self._freeze_layers(layers=freeze_layers)

    def forward(self, x: torch.Tensor, wavelengths: torch.Tensor, drop_masks=None,
                aux_drop_mask: torch.Tensor | None = None, lowres_logits: bool = False) -> SegmentationOutput:
        """dofa.py:83-107.  ``drop_masks`` / ``aux_drop_mask`` pin the stochastic draws (tests).  ``lowres_logits`` (not in the
        reference; off by default): ``out`` / ``aux`` come back as ``gdlhip.nn.LowresLogits`` -- the two heads' own maps plus the
        size the final ``F.interpolate`` would give them -- for a training step whose loss (``gdlhip.nn.DiceLoss``) evaluates the
        resize on the fly.  It travels through ``DistributedDataParallel.forward`` as a keyword argument."""
        image_size = x.shape[2:]
        with gnn.counter_batch():       # the BatchNorm step counters of the pass advance in one launch
            with rng("encoder"):        # (roctx ranges for rocprofv3 --marker-trace: GDL_ROCTX=1, gdlhip/markers.py)
                taps = self.encoder.forward_features_nhwc(x, wavelengths, drop_masks)
            with rng("neck"):
                feats = self.neck.forward_nhwc(taps)
            with rng("decoder"):
                dec = self.decoder.forward_nhwc(feats)
            with rng("heads"):
                out = self.head.forward_logits(dec, image_size, lowres=lowres_logits)
                aux = self.aux_head.forward_logits(feats[-1], image_size, aux_drop_mask, lowres=lowres_logits)
        return self.output_struct(out=out, aux=aux)
