"""Segmentation UNet++ task (drop-in for the reference's tasks_with_models/segmentation_unetplus.py:33-400:
same constructor keywords and step hooks; the ``smp.UnetPlusPlus`` it instantiates at :126-131 is replaced by the
HIP-kernel model in geo_deep_learning.models.segmentation.unetplusplus)."""

from __future__ import annotations

from collections.abc import Callable
from typing import Any

import torch
from torch import Tensor

from geo_deep_learning.models.segmentation.unetplusplus import UnetPlusPlus
from geo_deep_learning.tasks_with_models.segmentation_dofa import LightningModule
from geo_deep_learning.utils.models import load_weights_from_checkpoint
from gdlhip import nn as gnn


class SegmentationUnetPlus(LightningModule):
    """segmentation_unetplus.py:33-83 (constructor), :124-142 (configure_model), :223-283 (steps)."""

    def __init__(self, encoder: str, image_size: tuple[int, int], in_channels: int, num_classes: int,
                 max_samples: int, loss: Callable, optimizer: Callable = torch.optim.Adam,
                 scheduler: Callable = torch.optim.lr_scheduler.ConstantLR,
                 scheduler_config: dict[str, Any] | None = None, weights: str | None = None,
                 class_labels: list[str] | None = None, class_colors: list[str] | None = None,
                 weights_from_checkpoint_path: str | None = None, **kwargs: object) -> None:
        super().__init__()
        self.save_hyperparameters(encoder=encoder, in_channels=in_channels, num_classes=num_classes, **kwargs)
        self.encoder, self.in_channels, self.num_classes = encoder, in_channels, num_classes
        self.image_size, self.max_samples = tuple(image_size), max_samples
        self.loss = loss
        self.optimizer, self.scheduler = optimizer, scheduler
        self.scheduler_config = scheduler_config or {"interval": "epoch"}
        self.weights = weights
        self.weights_from_checkpoint_path = weights_from_checkpoint_path
        self.class_colors = class_colors
        self.threshold = 0.5
        n = num_classes + 1 if num_classes == 1 else num_classes
        self.labels = [str(i) for i in range(n)] if class_labels is None else class_labels

    def configure_model(self) -> None:
        if getattr(self, "model", None) is not None:
            return
        self.model = UnetPlusPlus(encoder_name=self.encoder, in_channels=self.in_channels,
                                  encoder_weights=self.weights, classes=self.num_classes)
        if self.weights_from_checkpoint_path:
            load_weights_from_checkpoint(self.model, self.weights_from_checkpoint_path,
                                         load_parts=self.hparams.get("load_parts"), map_location=self.device)

    def configure_optimizers(self):
        optimizer = self.optimizer(self.parameters())
        return [optimizer], [{"scheduler": self.scheduler(optimizer), **self.scheduler_config}]

    def forward(self, image: Tensor) -> Tensor:
        return self.model(image)

    def training_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        y_hat = self(batch["image"])
        loss = self.loss(y_hat, batch["mask"])      # the reference passes the [B,1,H,W] mask as is (:232)
        self.log("train_loss", loss, batch_size=batch["image"].shape[0], on_step=False, on_epoch=True, sync_dist=True)
        return loss

    def _apply_aug(self):
        """The reference's kornia pipeline (segmentation_unetplus.py:84-122) as one GPU kernel (gdlhip.augment)."""
        from gdlhip.augment import reference_pipeline
        return reference_pipeline(tuple(self.image_size))

    def on_after_batch_transfer(self, batch: dict[str, Any], dataloader_idx: int) -> dict[str, Any]:  # noqa: ARG002
        """The reference augments on the CPU in ``on_before_batch_transfer``; here the batch is augmented on the GPU
        right after the transfer (training only)."""
        trainer = getattr(self, "trainer", None)
        if trainer is not None and getattr(trainer, "training", False) and batch["image"].is_cuda:
            batch = self._apply_aug()(batch)
        return batch

    def validation_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        y_hat = self(batch["image"])
        loss = self.loss(y_hat, batch["mask"])
        self.log("val_loss", loss, batch_size=batch["image"].shape[0], on_step=False, on_epoch=True, sync_dist=True)
        if self.num_classes == 1:
            return (y_hat.sigmoid().squeeze(1) > self.threshold).long()
        return gnn.predict_mask(y_hat)
