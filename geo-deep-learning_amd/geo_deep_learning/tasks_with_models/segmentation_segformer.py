"""Segmentation SegFormer task (drop-in for the reference's tasks_with_models/segmentation_segformer.py:32-384)."""

from __future__ import annotations

from collections.abc import Callable
from typing import Any

import torch
from torch import Tensor

from geo_deep_learning.models.segmentation.segformer import SegFormerSegmentationModel
from geo_deep_learning.tasks_with_models._common import LightningModule, SegmentationTaskHooks
from geo_deep_learning.utils.models import load_weights_from_checkpoint


class SegmentationSegformer(SegmentationTaskHooks, LightningModule):
    """Same constructor keywords / hooks as the reference (segmentation_segformer.py:35-54,128-316)."""

    def __init__(self, encoder: str, in_channels: int, num_classes: int, max_samples: int, loss: Callable,
                 image_size: tuple[int, int] = (512, 512), weights: str | None = None,
                 optimizer: Callable = torch.optim.Adam,
                 scheduler: Callable = torch.optim.lr_scheduler.ConstantLR,
                 scheduler_config: dict[str, Any] | None = None, freeze_layers: list[str] | None = None,
                 class_labels: list[str] | None = None, class_colors: list[str] | None = None,
                 weights_from_checkpoint_path: str | None = None, *, use_dynamic_encoder: bool = False,
                 **kwargs: object) -> None:
        super().__init__()
        try:
            self.save_hyperparameters(encoder=encoder, in_channels=in_channels, num_classes=num_classes, **kwargs)
        except TypeError:  # real Lightning inspects the frame instead of taking kwargs
            self.save_hyperparameters()
        self.encoder, self.in_channels, self.weights = encoder, in_channels, weights
        self.image_size = tuple(image_size)
        self.use_dynamic_encoder = use_dynamic_encoder
        self.freeze_layers = freeze_layers
        self.weights_from_checkpoint_path = weights_from_checkpoint_path
        self.optimizer, self.scheduler = optimizer, scheduler
        self.scheduler_config = scheduler_config or {"interval": "epoch"}
        self.class_colors, self.max_samples, self.num_classes = class_colors, max_samples, num_classes
        self.loss = loss
        self._init_metrics(num_classes, class_labels)

    def configure_model(self) -> None:
        """segmentation_segformer.py:128-148."""
        if getattr(self, "model", None) is not None:
            return
        self.model = SegFormerSegmentationModel(encoder=self.encoder, in_channels=self.in_channels,
                                                weights=self.weights, freeze_layers=self.freeze_layers,
                                                num_classes=self.num_classes,
                                                use_dynamic_encoder=self.use_dynamic_encoder)
        if self.weights_from_checkpoint_path:
            load_weights_from_checkpoint(self.model, self.weights_from_checkpoint_path,
                                         load_parts=self.hparams.get("load_parts"), map_location=self.device)

    def forward(self, image: Tensor) -> Tensor:
        return self.model(image)

    def _loss(self, batch: dict[str, Any]):
        x = batch["image"]
        y = batch["mask"].squeeze(1).long()
        y_hat = self(x)
        return y_hat, y, self.loss(y_hat, y), x.shape[0]

    def training_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        """segmentation_segformer.py:216-241."""
        _, _, loss, bs = self._loss(batch)
        self.train_samples_count += bs
        self._log_loss("train_loss", loss, bs)
        return loss

    def validation_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        """segmentation_segformer.py:243-274."""
        y_hat, _, loss, bs = self._loss(batch)
        self.val_samples_count += bs
        self._log_loss("val_loss", loss, bs)
        return self._predict(y_hat)

    def test_step(self, batch: dict[str, Any], batch_idx: int) -> None:  # noqa: ARG002
        """segmentation_segformer.py:276-316."""
        y_hat, y, loss, bs = self._loss(batch)
        self.test_samples_count += bs
        self._log_test_metrics(self._predict(y_hat), y, loss, bs)
