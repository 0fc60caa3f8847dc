"""Segmentation UNet++ task (drop-in for the reference's tasks_with_models/segmentation_unetplus.py:33-400:
same constructor keywords and step hooks; the ``smp.UnetPlusPlus`` it instantiates at :126-131 is replaced by the
HIP-kernel model in geo_deep_learning.models.segmentation.unetplusplus)."""

from __future__ import annotations

from collections.abc import Callable
from typing import Any

import torch
from torch import Tensor

from geo_deep_learning.models.segmentation.unetplusplus import UnetPlusPlus
from geo_deep_learning.tasks_with_models._common import LightningModule, SegmentationTaskHooks
from geo_deep_learning.utils.models import load_weights_from_checkpoint


class SegmentationUnetPlus(SegmentationTaskHooks, LightningModule):
    """segmentation_unetplus.py:33-83 (constructor), :124-142 (configure_model), :223-320 (steps)."""

    def __init__(self, encoder: str, image_size: tuple[int, int], in_channels: int, num_classes: int,
                 max_samples: int, loss: Callable, optimizer: Callable = torch.optim.Adam,
                 scheduler: Callable = torch.optim.lr_scheduler.ConstantLR,
                 scheduler_config: dict[str, Any] | None = None, weights: str | None = None,
                 class_labels: list[str] | None = None, class_colors: list[str] | None = None,
                 weights_from_checkpoint_path: str | None = None, **kwargs: object) -> None:
        super().__init__()
        try:
            self.save_hyperparameters(encoder=encoder, in_channels=in_channels, num_classes=num_classes, **kwargs)
        except TypeError:  # real Lightning inspects the frame instead of taking kwargs
            self.save_hyperparameters()
        self.encoder, self.in_channels, self.num_classes = encoder, in_channels, num_classes
        self.image_size, self.max_samples = tuple(image_size), max_samples
        self.loss = loss
        self.optimizer, self.scheduler = optimizer, scheduler
        self.scheduler_config = scheduler_config or {"interval": "epoch"}
        self.weights = weights
        self.weights_from_checkpoint_path = weights_from_checkpoint_path
        self.class_colors = class_colors
        self._init_metrics(num_classes, class_labels)

    def configure_model(self) -> None:
        if getattr(self, "model", None) is not None:
            return
        self.model = UnetPlusPlus(encoder_name=self.encoder, in_channels=self.in_channels,
                                  encoder_weights=self.weights, classes=self.num_classes)
        if self.weights_from_checkpoint_path:
            load_weights_from_checkpoint(self.model, self.weights_from_checkpoint_path,
                                         load_parts=self.hparams.get("load_parts"), map_location=self.device)

    def forward(self, image: Tensor) -> Tensor:
        return self.model(image)

    def training_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        """segmentation_unetplus.py:223-247: the ``[B,1,H,W]`` mask goes to the loss AS IS (the squeeze at :232 is
        commented out in the reference); smp's DiceLoss views it ``[B,-1]`` / ``[B,1,-1]`` itself."""
        x, y = batch["image"], batch["mask"]
        loss = self.loss(self(x), y)
        self.train_samples_count += x.shape[0]
        self._log_loss("train_loss", loss, x.shape[0])
        return loss

    def validation_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        """segmentation_unetplus.py:249-276."""
        x, y = batch["image"], batch["mask"]
        y_hat = self(x)
        self.val_samples_count += x.shape[0]
        self._log_loss("val_loss", self.loss(y_hat, y), x.shape[0])
        return self._predict(y_hat)

    def test_step(self, batch: dict[str, Any], batch_idx: int) -> None:  # noqa: ARG002
        """segmentation_unetplus.py:278-320: loss on the un-squeezed mask, metrics on ``mask.squeeze(1).long()``."""
        x, y = batch["image"], batch["mask"]
        y_hat = self(x)
        loss = self.loss(y_hat, y)
        self.test_samples_count += x.shape[0]
        self._log_test_metrics(self._predict(y_hat), y.squeeze(1).long(), loss, x.shape[0])
