"""Oracle model assembly, loss, preprocess, procedural weights, synthetic inputs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""

from __future__ import annotations

import zlib
from typing import NamedTuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .decoder import FCNHead, MultiLevelNeck, SegmentationHead, UperNetDecoder
from .encoder import create_dofa_base, create_dofa_large


class SegmentationOutput(NamedTuple):
    """Reference: models/heads/segmentation_head.py:9-13."""

    out: Tensor
    aux: Tensor | None


class DOFASegmentationModel(nn.Module):
    """Reference: models/segmentation/dofa.py:24-107."""

    def __init__(self, encoder: str = "dofa_base", image_size=(512, 512),
                 freeze_layers: list[str] | None = None, num_classes: int = 1,
                 *, pretrained: bool = False, _encoder_kwargs: dict | None = None) -> None:
        super().__init__()
        if pretrained:
            msg = "oracle: pretrained download is out of scope (no network)"
            raise ValueError(msg)
        kw = _encoder_kwargs or {}
        if encoder == "dofa_base":
            self.encoder = create_dofa_base(img_size=image_size, **kw)
        elif encoder == "dofa_large":
            self.encoder = create_dofa_large(img_size=image_size, **kw)
        elif encoder == "dofa_tiny_test":  # small-shape fixture config (not in the reference)
            from .encoder import DOFAv2
            self.encoder = DOFAv2(img_size=image_size, **kw)
        else:
            msg = f"Invalid encoder: {encoder}"
            raise ValueError(msg)
        self.embed_dim = self.encoder.embed_dim
        e = self.embed_dim
        self.neck = MultiLevelNeck([e] * 4, [e] * 4, scales=[4, 2, 1, 0.5])
        self.decoder = UperNetDecoder([e] * 4, (1, 2, 3, 6), channels=256, align_corners=False)
        self.aux_head = FCNHead(e, channels=256, num_classes=num_classes)
        self.head = SegmentationHead(256, num_classes)
        if freeze_layers:
            # models/segmentation/base.py:40-44: substring match on parameter names
            for name, p in self.named_parameters():
                if any(layer in name for layer in freeze_layers):
                    p.requires_grad = False

    def forward(self, x: Tensor, wavelengths: Tensor, drop_masks=None,
                aux_drop_mask: Tensor | None = None) -> SegmentationOutput:
        size = x.shape[2:]
        feats = self.neck(self.encoder(x, wavelengths, drop_masks))
        out = self.head(self.decoder(feats))
        out = F.interpolate(out, size=size, mode="bilinear", align_corners=False)
        aux = self.aux_head(feats[-1], aux_drop_mask)
        aux = F.interpolate(aux, size=size, mode="bilinear", align_corners=False)
        return SegmentationOutput(out=out, aux=aux)


def dice_loss_multiclass(logits: Tensor, target: Tensor, smooth: float = 0.0,
                         eps: float = 1e-7) -> Tensor:
    """smp 0.5.0 ``DiceLoss(mode="multiclass")`` (third-party).

    Configured at configs/dofa_config_RGB.yaml:58-61.  "parity unpinned".
    """
    b, c = logits.shape[:2]
    p = logits.log_softmax(dim=1).exp().view(b, c, -1)
    y = F.one_hot(target.view(b, -1), c).permute(0, 2, 1).type_as(p)
    inter = torch.sum(p * y, dim=(0, 2))
    card = torch.sum(p + y, dim=(0, 2))
    dice = (2.0 * inter + smooth) / (card + smooth).clamp_min(eps)
    loss = (1.0 - dice) * (y.sum(dim=(0, 2)) > 0).to(p.dtype)
    return loss.mean()


def dice_loss_binary(logits: Tensor, target: Tensor, smooth: float = 0.0, eps: float = 1e-7) -> Tensor:
    """smp 0.5.0 ``DiceLoss(mode="binary")`` (third-party, absent from /root/reference; restated from its published
    algorithm, "parity unpinned"): ``p = logsigmoid(x).exp()``, both tensors viewed ``[B, 1, -1]``, soft Dice over dims
    (0, 2), ``loss = (1 - dice) * [sum y > 0]``, mean over the single class.  Used by the reference's UNet++ config
    (configs/unetplus_config_RGB.yaml:36-47: ``num_classes: 1``, ``mode: "binary"``); the task passes the
    ``[B,1,H,W]`` mask unsqueezed (segmentation_unetplus.py:232)."""
    b = target.shape[0]
    p = F.logsigmoid(logits).exp().view(b, 1, -1)
    y = target.view(b, 1, -1).type_as(p)
    inter = torch.sum(p * y, dim=(0, 2))
    card = torch.sum(p + y, dim=(0, 2))
    dice = (2.0 * inter + smooth) / (card + smooth).clamp_min(eps)
    loss = (1.0 - dice) * (y.sum(dim=(0, 2)) > 0).to(p.dtype)
    return loss.mean()


def training_loss(outputs: SegmentationOutput, mask: Tensor) -> Tensor:
    """loss_main + 0.4*loss_aux.  Reference: segmentation_dofa.py:224-228."""
    y = mask.squeeze(1).long()
    return dice_loss_multiclass(outputs.out, y) + 0.4 * dice_loss_multiclass(outputs.aux, y)


def predict_mask(outputs: SegmentationOutput) -> Tensor:
    """softmax(dim=1).argmax(dim=1).  Reference: segmentation_dofa.py:281."""
    return outputs.out.softmax(dim=1).argmax(dim=1)


def normalization(x: Tensor, image_min=0, image_max=255, norm_min=0.0, norm_max=1.0) -> Tensor:
    """Reference: utils/tensors.py:10-22."""
    shape = x.shape
    x = (norm_max - norm_min) * (x - image_min) / (image_max - image_min) + norm_min
    return x.reshape(shape)


def standardization(x: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Reference: utils/tensors.py:25-35 (mean/std broadcast as [C,1] over [B,C,HW])."""
    shape = x.shape
    b, c = x.shape[:2]
    x = x.reshape(b, c, -1)
    return ((x - mean) / std).reshape(shape)


# --------------------------------------------------------------------------
# Portable procedural weights + synthetic inputs (numpy PCG64; no torch RNG).
# --------------------------------------------------------------------------
def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def procedural_state_dict(model: nn.Module, seed: int = 42) -> dict[str, Tensor]:
    """Deterministic, non-degenerate values for every state-dict entry."""
    out = {}
    for key, t in model.state_dict().items():
        g = _rng(seed, key)
        shape = tuple(t.shape)
        leaf = key.rsplit(".", 1)[-1]
        parent = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros_like(t)
            continue
        if key.endswith("pos_embed"):
            out[key] = t.clone()  # fixed 2-D sincos (dofa_v2.py:268-276)
            continue
        is_norm = parent.startswith("norm")
        if leaf == "running_var":
            v = g.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            v = g.normal(0.0, 0.1, shape)
        elif leaf == "weight" and (is_norm or len(shape) == 1):   # every 1-D "weight" is a norm scale
            v = g.uniform(0.8, 1.2, shape)
        elif leaf == "gamma":
            v = g.uniform(0.05, 0.2, shape) * np.where(g.uniform(size=shape) < 0.5, -1.0, 1.0)
        elif leaf in ("cls_token", "weight_tokens", "bias_token"):
            v = g.normal(0.0, 0.2, shape)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = g.normal(0.0, (2.0 / fan_in) ** 0.5, shape)
        else:
            v = g.normal(0.0, 0.05, shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(t.dtype)
    return out


RGB_MEAN = [0.3992, 0.4283, 0.3998]   # configs/dofa_config_RGB.yaml:91-94
RGB_STD = [0.1672, 0.1800, 0.1584]    # configs/dofa_config_RGB.yaml:95-98
WAVELENGTHS = {
    3: [0.665, 0.549, 0.481],          # configs/dofa_config_RGB.yaml:50
    6: [0.665, 0.549, 0.481, 0.842, 1.610, 2.190],
    10: [0.490, 0.560, 0.665, 0.705, 0.740, 0.783, 0.842, 0.865, 1.610, 2.190],
}


def synthetic_batch(batch: int, bands: int = 3, size: int = 512, num_classes: int = 5,
                    seed: int = 42) -> dict[str, Tensor]:
    """SURVEY section 8(d) synthetic inputs: uint8 tile -> /255 -> standardise."""
    g = np.random.default_rng([seed, batch, bands, size])
    u8 = torch.from_numpy(g.integers(0, 256, (batch, bands, size, size), dtype=np.uint8))
    mask = torch.from_numpy(g.integers(0, num_classes, (batch, 1, size, size), dtype=np.int64))
    mean = torch.tensor([RGB_MEAN[i % 3] for i in range(bands)]).view(-1, 1)
    std = torch.tensor([RGB_STD[i % 3] for i in range(bands)]).view(-1, 1)
    image = standardization(normalization(u8.float()), mean, std)
    wv = torch.tensor(WAVELENGTHS[bands], dtype=torch.float32)
    return {"image_u8": u8, "image": image, "mask": mask, "wavelengths": wv,
            "mean": mean.view(-1), "std": std.view(-1)}
