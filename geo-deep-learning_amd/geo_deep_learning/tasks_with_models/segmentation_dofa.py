"""Segmentation DOFA task (drop-in for the reference's tasks_with_models/segmentation_dofa.py).

Same class name, constructor keyword arguments and Lightning hooks as the reference
(/root/reference/geo_deep_learning/tasks_with_models/segmentation_dofa.py:33-338), so
``model.class_path: tasks_with_models.segmentation_dofa.SegmentationDOFA`` resolves unchanged.
``lightning`` is optional: when it is not installed the class derives from a minimal stand-in
(``gdlhip.trainer.LightningModule``) and ``gdlhip.trainer.MiniTrainer`` drives the same hooks.

Host logic only; all tensor arithmetic (model, Dice loss, softmax->argmax, IoU counts via the
mask kernels) runs in libgdlhip.so.  MLflow figure logging of the reference is outside the hot path
 and is not rebuilt.
"""

from __future__ import annotations

import logging
from collections.abc import Callable
from typing import Any

import torch
from torch import Tensor

from geo_deep_learning.models.segmentation.dofa import DOFASegmentationModel
from geo_deep_learning.tasks_with_models._common import LightningModule, SegmentationTaskHooks
from geo_deep_learning.utils.models import load_weights_from_checkpoint
from gdlhip.markers import rng

logger = logging.getLogger(__name__)


class SegmentationDOFA(SegmentationTaskHooks, LightningModule):
    """Segmentation DOFA model (segmentation_dofa.py:33-414)."""

    def __init__(  # noqa: PLR0913
        self,
        encoder: str,
        *,
        pretrained: bool,
        image_size: tuple[int, int],
        num_classes: int,
        max_samples: int,
        loss: Callable,
        optimizer: Callable = torch.optim.Adam,
        scheduler: Callable = torch.optim.lr_scheduler.ConstantLR,
        scheduler_config: dict[str, Any] | None = None,
        freeze_layers: list[str] | None = None,
        class_labels: list[str] | None = None,
        class_colors: list[str] | None = None,
        weights_from_checkpoint_path: str | None = None,
        **kwargs: object,
    ) -> None:
        super().__init__()
        try:
            self.save_hyperparameters(encoder=encoder, pretrained=pretrained, image_size=image_size,
                                      num_classes=num_classes, freeze_layers=freeze_layers, **kwargs)
        except TypeError:  # real Lightning inspects the frame instead of taking kwargs
            self.save_hyperparameters()
        self.encoder = encoder
        self.pretrained = pretrained
        self.image_size = tuple(image_size)
        self.freeze_layers = freeze_layers
        self.weights_from_checkpoint_path = weights_from_checkpoint_path
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.scheduler_config = scheduler_config or {"interval": "epoch"}
        self.class_colors = class_colors
        self.max_samples = max_samples
        self.num_classes = num_classes
        self.loss = loss
        self._init_metrics(num_classes, class_labels)

    def configure_model(self) -> None:
        """segmentation_dofa.py:123-144."""
        if getattr(self, "model", None) is not None:
            return
        self.model = DOFASegmentationModel(encoder=self.encoder, image_size=self.image_size,
                                           freeze_layers=self.freeze_layers, num_classes=self.num_classes,
                                           pretrained=self.pretrained)
        if self.weights_from_checkpoint_path:
            load_weights_from_checkpoint(self.model, self.weights_from_checkpoint_path,
                                         load_parts=self.hparams.get("load_parts"), map_location=self.device)

    def forward(self, image: Tensor, wavelengths: Tensor) -> Tensor:
        return self.model(image, wavelengths)

    def _loss(self, batch: dict[str, Any], lowres_logits: bool = False):
        x, y, wv = batch["image"], batch["mask"], batch["wavelengths"]
        y = y.squeeze(1).long()
        # lowres_logits: the loss (and, in validation / test, the arg-max mask) is all the step needs from the logits, and gdlhip's DiceLoss evaluates it -- and its
        # gradient -- from the heads' own maps (128 x 128 main, 16 x 16 auxiliary); the reference's F.interpolate to 512 x 512
        # (dofa.py:89-105) and the 168 MB tensor it produces per head exist only where something reads them (validation / test)
        outputs = self.model(x, wv, lowres_logits=True) if lowres_logits else self(x, wv)
        with rng("loss"):
            loss = self.loss(outputs.out, y) + 0.4 * self.loss(outputs.aux, y)
        return outputs, y, loss, x.shape[0]

    def training_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        """segmentation_dofa.py:213-241."""
        from gdlhip import nn as gnn
        fused = gnn.FUSE_LOWRES_DICE and isinstance(self.loss, gnn.DiceLoss) and self.loss.mode == "multiclass"
        _, _, loss, bs = self._loss(batch, lowres_logits=fused)
        self.train_samples_count += bs
        self._log_loss("train_loss", loss, bs)
        return loss

    def _lowres_eval(self) -> bool:
        """Validation / test need the two Dice terms and the arg-max mask of ``outputs.out``, nothing else of the logits: with gdlhip's
        multiclass DiceLoss both come straight from the heads' own maps (``gnn.DiceLoss`` / ``gnn.predict_mask`` on LowresLogits: the
        same values, bit for bit, as from the resized [B, K, H, W] tensors, which are then never written)."""
        from gdlhip import nn as gnn
        return (gnn.FUSE_LOWRES_DICE and isinstance(self.loss, gnn.DiceLoss) and self.loss.mode == "multiclass"
                and self.num_classes > 1)

    def validation_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        """segmentation_dofa.py:251-283."""
        outputs, _, loss, bs = self._loss(batch, lowres_logits=self._lowres_eval())
        self.val_samples_count += bs
        self._log_loss("val_loss", loss, bs)
        return self._predict(outputs.out)

    def test_step(self, batch: dict[str, Any], batch_idx: int) -> None:  # noqa: ARG002
        """segmentation_dofa.py:293-338 (per-class IoU; figure logging not rebuilt)."""
        outputs, y, loss, bs = self._loss(batch, lowres_logits=self._lowres_eval())
        self.test_samples_count += bs
        self._log_test_metrics(self._predict(outputs.out), y, loss, bs)
