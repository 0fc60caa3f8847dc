"""GPU-side training augmentation with the class names / arguments of the kornia pipeline the reference builds in
``_apply_aug`` (tasks_with_models/segmentation_dofa.py:91-121; same in the SegFormer and UNet++ tasks) and applies
on the CPU in ``on_before_batch_transfer`` (:201-211).

Per batch ``AugmentationSequential(random_apply=1)`` picks ONE of its augmentations; that augmentation then draws,
per sample, whether it applies (``p``) and its parameters on the host (torch's CPU generator: ``torch.manual_seed``
governs it); the pixels move in ONE HIP kernel (``gdl_augment``), optionally fused with ``/255 -> (x-mean)/std`` when
the tile is still raw.  kornia 0.8.2 is third-party and absent from the image: the parameter distributions are
restated from its documented behaviour ("parity unpinned"); the resampling itself is checked against
``F.interpolate`` / ``torch.flip`` / ``torch.rot90`` in tests/test_input_stage.py.
"""

from __future__ import annotations

import math

import torch
from torch import Tensor

from . import ops

NONE, HFLIP, VFLIP, ROT90, CROP = 0, 1, 2, 3, 4


class _Aug:
    def __init__(self, p: float = 0.5, **_kw: object) -> None:
        self.p = float(p)

    def sample(self, b: int, h: int, w: int) -> Tensor:
        """f32 [b, 8] parameter rows {kind, k, y0, x0, h, w, 0, 0}; kind 0 where the Bernoulli(p) draw fails."""
        prm = torch.zeros((b, 8), dtype=torch.float32)
        apply = torch.rand(b) < self.p
        for i in range(b):
            if apply[i]:
                self._fill(prm[i], h, w)
        return prm

    def _fill(self, row: Tensor, h: int, w: int) -> None:
        raise NotImplementedError


class RandomHorizontalFlip(_Aug):
    def _fill(self, row: Tensor, h: int, w: int) -> None:
        row[0] = HFLIP


class RandomVerticalFlip(_Aug):
    def _fill(self, row: Tensor, h: int, w: int) -> None:
        row[0] = VFLIP


class RandomRotation90(_Aug):
    """Quarter turns, ``times=(lo, hi)`` inclusive, anticlockwise like ``torch.rot90``."""

    def __init__(self, times: tuple[int, int] = (1, 3), p: float = 0.5, **kw: object) -> None:
        super().__init__(p)
        self.times = (int(times[0]), int(times[1]))

    def _fill(self, row: Tensor, h: int, w: int) -> None:
        if h != w:
            msg = "RandomRotation90 needs square tiles"
            raise ValueError(msg)
        row[0] = ROT90
        row[1] = int(torch.randint(self.times[0], self.times[1] + 1, (1,)))


class RandomResizedCrop(_Aug):
    """Crop ``scale`` x area with aspect in ``ratio`` (10 tries, then the central fallback), resized to ``size``
    (bilinear / nearest-for-masks, align_corners=False)."""

    def __init__(self, size: tuple[int, int], scale: tuple[float, float] = (0.08, 1.0),
                 ratio: tuple[float, float] = (3.0 / 4.0, 4.0 / 3.0), p: float = 0.5, **kw: object) -> None:
        super().__init__(p)
        self.size, self.scale, self.ratio = tuple(size), scale, ratio

    def _fill(self, row: Tensor, h: int, w: int) -> None:
        if self.size != (h, w):
            msg = f"RandomResizedCrop: output size {self.size} must equal the tile size {(h, w)}"
            raise ValueError(msg)
        area = h * w
        log_r = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * float(torch.empty(1).uniform_(self.scale[0], self.scale[1]))
            aspect = math.exp(float(torch.empty(1).uniform_(log_r[0], log_r[1])))
            cw, ch = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if 0 < cw <= w and 0 < ch <= h:
                y0 = int(torch.randint(0, h - ch + 1, (1,)))
                x0 = int(torch.randint(0, w - cw + 1, (1,)))
                row[0], row[2], row[3], row[4], row[5] = CROP, y0, x0, ch, cw
                return
        in_ratio = w / h
        if in_ratio < min(self.ratio):
            cw, ch = w, int(round(w / min(self.ratio)))
        elif in_ratio > max(self.ratio):
            ch, cw = h, int(round(h * max(self.ratio)))
        else:
            cw, ch = w, h
        row[0], row[2], row[3], row[4], row[5] = CROP, (h - ch) // 2, (w - cw) // 2, ch, cw


class AugmentationSequential:
    """``aug({"image": x, "mask": y})`` -> the same dict, augmented (kornia.augmentation.AugmentationSequential with
    ``data_keys=None``); ``random_apply=k`` draws k of the members per call, ``False`` applies all in order."""

    def __init__(self, *augs: _Aug, data_keys: object = None, random_apply: int | bool = False) -> None:
        self.augs = list(augs)
        self.random_apply = random_apply

    def _pick(self) -> list[_Aug]:
        if not self.random_apply:
            return self.augs
        k = int(self.random_apply)
        idx = sorted(torch.randperm(len(self.augs))[:k].tolist())
        return [self.augs[i] for i in idx]

    def __call__(self, batch: dict, mean: Tensor | None = None, std: Tensor | None = None) -> dict:
        img, mask = batch["image"], batch.get("mask")
        b, _, h, w = img.shape
        out = dict(batch)
        picked = self._pick()
        if not picked and mean is not None:
            picked = [_Identity()]
        for i, aug in enumerate(picked):
            prm = aug.sample(b, h, w).to(img.device, non_blocking=True)
            m = None if mask is None else mask.reshape(b, h, w).long()
            img, om = ops.augment(img.contiguous(), m, prm, mean if i == 0 else None, std if i == 0 else None)
            if om is not None:
                mask = om.view(batch["mask"].shape)
        out["image"] = img
        if mask is not None:
            out["mask"] = mask
        return out


class _Identity(_Aug):
    def __init__(self) -> None:
        super().__init__(0.0)

    def _fill(self, row: Tensor, h: int, w: int) -> None:
        return


def reference_pipeline(image_size: tuple[int, int]) -> AugmentationSequential:
    """The five-member, one-of-five pipeline of segmentation_dofa.py:91-121."""
    return AugmentationSequential(
        RandomHorizontalFlip(p=0.5), RandomVerticalFlip(p=0.5), RandomRotation90(times=(1, 3), p=0.5),
        RandomResizedCrop(size=image_size, scale=(1.0, 2.0), p=0.5),
        RandomResizedCrop(size=image_size, scale=(0.5, 1.0), p=0.5), data_keys=None, random_apply=1)
