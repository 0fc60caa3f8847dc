"""Segmentation DOFA task (drop-in for the reference's tasks_with_models/segmentation_dofa.py).

Same class name, constructor keyword arguments and Lightning hooks as the reference
(/root/reference/geo_deep_learning/tasks_with_models/segmentation_dofa.py:33-338), so
``model.class_path: tasks_with_models.segmentation_dofa.SegmentationDOFA`` resolves unchanged.
``lightning`` is optional: when it is not installed the class derives from a minimal stand-in
that provides the attributes the hooks touch (``self.log``, ``hparams``, ``trainer``).

Host logic only; all tensor arithmetic (model, Dice loss, softmax->argmax, IoU counts via the
mask kernels) runs in libgdlhip.so.  The kornia augmentation hook / MLflow figure logging of
the reference are outside the hot path and are not rebuilt.
"""

from __future__ import annotations

import logging
import math
from collections.abc import Callable
from typing import Any

import torch
from torch import Tensor

from geo_deep_learning.models.segmentation.dofa import DOFASegmentationModel
from geo_deep_learning.utils.models import load_weights_from_checkpoint
from gdlhip import nn as gnn
from gdlhip.metrics import ClasswiseWrapper, MeanIoU

try:  # pragma: no cover - lightning is absent in the build image
    from lightning.pytorch import LightningModule
except ImportError:  # minimal stand-in with the surface the hooks below use
    class LightningModule(torch.nn.Module):
        def __init__(self) -> None:
            super().__init__()
            self.hparams: dict[str, Any] = {}
            self.trainer = None
            self.logged: dict[str, Any] = {}

        def save_hyperparameters(self, **kw: Any) -> None:
            self.hparams.update(kw)

        def log(self, name: str, value: Any, **_kw: Any) -> None:
            self.logged[name] = value

        def log_dict(self, d: dict[str, Any], **_kw: Any) -> None:
            self.logged.update(d)

        @property
        def device(self) -> torch.device:
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

logger = logging.getLogger(__name__)


class SegmentationDOFA(LightningModule):
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
        if hasattr(self, "save_hyperparameters"):
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
        self.threshold = 0.5
        self.loss = loss
        n = num_classes + 1 if num_classes == 1 else num_classes
        self.labels = [str(i) for i in range(n)] if class_labels is None else class_labels
        self._iou_classes = n
        self.iou_metric = MeanIoU(num_classes=n, per_class=True, input_format="index", include_background=True)
        self.iou_classwise_metric = ClasswiseWrapper(self.iou_metric, labels=self.labels)
        self.train_samples_count = 0
        self.val_samples_count = 0
        self.test_samples_count = 0

    # ------------------------------------------------------------------ Lightning hooks
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

    def configure_optimizers(self):
        """segmentation_dofa.py:146-195 (OneCycleLR special case reduced to total_steps)."""
        optimizer = self.optimizer(self.parameters())
        sched_cfg = self.hparams.get("scheduler") if isinstance(self.hparams.get("scheduler"), dict) else None
        if sched_cfg and sched_cfg.get("class_path") == "torch.optim.lr_scheduler.OneCycleLR":
            init = sched_cfg.get("init_args", {})
            steps = getattr(self.trainer, "estimated_stepping_batches", -1) if self.trainer else -1
            if steps is None or steps <= 0:
                dm = getattr(self.trainer, "datamodule", None) if self.trainer else None
                if dm is not None and getattr(dm, "epoch_size", None) is not None:
                    per_epoch = math.ceil(dm.epoch_size / (dm.batch_size * self.trainer.accumulate_grad_batches))
                    steps = per_epoch * self.trainer.max_epochs
                else:
                    steps = init.get("total_steps")
            scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=init.get("max_lr"),
                                                            total_steps=steps)
        else:
            scheduler = self.scheduler(optimizer)
        return [optimizer], [{"scheduler": scheduler, **self.scheduler_config}]

    def forward(self, image: Tensor, wavelengths: Tensor) -> Tensor:
        return self.model(image, wavelengths)

    def on_before_batch_transfer(self, batch: dict[str, Any], dataloader_idx: int) -> dict[str, Any]:  # noqa: ARG002
        """The reference runs a kornia augmentation pipeline on the host here
        (segmentation_dofa.py:201-211); SURVEY.md 8(f) ranks a GPU-side version as the next row."""
        return batch

    def _loss(self, batch: dict[str, Any]):
        x, y, wv = batch["image"], batch["mask"], batch["wavelengths"]
        y = y.squeeze(1).long()
        outputs = self(x, wv)
        loss = self.loss(outputs.out, y) + 0.4 * self.loss(outputs.aux, y)
        return outputs, y, loss, x.shape[0]

    def training_step(self, batch: dict[str, Any], batch_idx: int) -> Tensor:  # noqa: ARG002
        """segmentation_dofa.py:213-241."""
        _, _, loss, bs = self._loss(batch)
        self.train_samples_count += bs
        self.log("train_loss", loss, batch_size=bs, prog_bar=True, logger=True, on_step=False, on_epoch=True,
                 sync_dist=True, rank_zero_only=True)
        return loss

    def _predict(self, outputs) -> Tensor:
        if self.num_classes == 1:
            return (outputs.out.sigmoid().squeeze(1) > self.threshold).long()
        return gnn.predict_mask(outputs.out)  # softmax(dim=1).argmax(dim=1), segmentation_dofa.py:281

    def _apply_aug(self):
        """The reference's kornia pipeline (segmentation_dofa.py:91-121,201-211) as one GPU kernel (gdlhip.augment)."""
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
        """segmentation_dofa.py:251-283."""
        outputs, _, loss, bs = self._loss(batch)
        self.val_samples_count += bs
        self.log("val_loss", loss, batch_size=bs, prog_bar=True, logger=True, on_step=False, on_epoch=True,
                 sync_dist=True, rank_zero_only=True)
        return self._predict(outputs)

    def test_step(self, batch: dict[str, Any], batch_idx: int) -> None:  # noqa: ARG002
        """segmentation_dofa.py:293-338 (per-class IoU; figure logging not rebuilt)."""
        outputs, y, loss, bs = self._loss(batch)
        self.test_samples_count += bs
        y_hat = self._predict(outputs)
        metrics = self.iou_classwise_metric(y_hat, y)      # per-class IoU from the integer count kernel
        self.iou_classwise_metric.reset()
        metrics["test_loss"] = loss
        self.log_dict(metrics, batch_size=bs, prog_bar=False, logger=True, on_step=False, sync_dist=True,
                      rank_zero_only=True)

    def on_train_epoch_end(self) -> None:
        logger.info("Training epoch complete. Processed %d samples", self.train_samples_count)
        self.train_samples_count = 0

    def on_validation_epoch_end(self) -> None:
        logger.info("Validation epoch complete. Processed %d samples", self.val_samples_count)
        self.val_samples_count = 0

    def on_test_epoch_end(self) -> None:
        logger.info("Test epoch complete. Processed %d samples", self.test_samples_count)
        self.test_samples_count = 0
