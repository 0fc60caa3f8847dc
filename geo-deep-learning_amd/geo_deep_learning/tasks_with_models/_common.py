"""Hook code shared by the three task classes.  The reference repeats these methods verbatim in
tasks_with_models/segmentation_{dofa,segformer,unetplus}.py (``configure_optimizers`` :146-195 / :150-199 / :158-208,
``_apply_aug`` :91-121, the IoU metrics of ``__init__`` :71-85, the test-step tail :313-338, the epoch-end hooks); the
mirrors inherit them from one place.  Host logic only."""

from __future__ import annotations

import logging
import math
from typing import Any

import torch
from torch import Tensor

from gdlhip import nn as gnn
from gdlhip.metrics import ClasswiseWrapper, MeanIoU
from gdlhip.trainer import LightningModule  # noqa: F401  (lightning's class when installed, else the stand-in)

logger = logging.getLogger(__name__)


class SegmentationTaskHooks:
    """Mixin: expects ``self.num_classes``, ``self.image_size``, ``self.optimizer``, ``self.scheduler``,
    ``self.scheduler_config``, ``self.hparams``, ``self.trainer`` (the attributes the reference's classes set)."""

    threshold = 0.5

    # ------------------------------------------------------------------ metrics (segmentation_dofa.py:71-85)
    def _init_metrics(self, num_classes: int, class_labels: list[str] | None) -> None:
        n = num_classes + 1 if num_classes == 1 else num_classes
        self.labels = [str(i) for i in range(n)] if class_labels is None else class_labels
        self.iou_metric = MeanIoU(num_classes=n, per_class=True, input_format="index", include_background=True)
        self.iou_classwise_metric = ClasswiseWrapper(self.iou_metric, labels=self.labels)
        self._total_samples_visualized = 0
        self.train_samples_count = self.val_samples_count = self.test_samples_count = 0

    # ------------------------------------------------------------------ optimizers (segmentation_dofa.py:146-195)
    def configure_optimizers(self):
        optimizer = self.optimizer(self.parameters())
        sched_cfg = self.hparams.get("scheduler") if isinstance(self.hparams.get("scheduler"), dict) else {}
        if sched_cfg.get("class_path") == "torch.optim.lr_scheduler.OneCycleLR":
            init = sched_cfg.get("init_args", {}) or {}
            max_lr = init.get("max_lr")
            stepping_batches = getattr(self.trainer, "estimated_stepping_batches", -1)
            dm = getattr(self.trainer, "datamodule", None)
            if stepping_batches is not None and stepping_batches > -1:
                scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=max_lr, total_steps=stepping_batches)
            elif getattr(dm, "epoch_size", None) is not None:
                acc = self.trainer.accumulate_grad_batches
                steps_per_epoch = math.ceil(dm.epoch_size / (dm.batch_size * acc))
                buffer_steps = int(steps_per_epoch * acc)    # head-room against "Tried to step N times" (:177-183)
                scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=max_lr,
                                                                steps_per_epoch=steps_per_epoch + buffer_steps,
                                                                epochs=self.trainer.max_epochs)
            else:
                scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=max_lr,
                                                                total_steps=init.get("total_steps"))
        else:
            scheduler = self.scheduler(optimizer)
        return [optimizer], [{"scheduler": scheduler, **self.scheduler_config}]

    # ------------------------------------------------------------------ augmentation
    def _apply_aug(self):
        """The reference's kornia pipeline (segmentation_dofa.py:91-121) as one GPU kernel (gdlhip.augment)."""
        from gdlhip.augment import reference_pipeline
        return reference_pipeline(tuple(self.image_size))

    def on_before_batch_transfer(self, batch: dict[str, Any], dataloader_idx: int) -> dict[str, Any]:  # noqa: ARG002
        """The reference augments here, on the host (segmentation_dofa.py:201-211); this build augments on the GPU in
        ``on_after_batch_transfer``, so the host batch passes through."""
        return batch

    def on_after_batch_transfer(self, batch: dict[str, Any], dataloader_idx: int) -> dict[str, Any]:  # noqa: ARG002
        trainer = getattr(self, "trainer", None)
        img = batch.get("image") if isinstance(batch, dict) else None
        if trainer is not None and getattr(trainer, "training", False) and isinstance(img, Tensor):
            if img.is_cuda:
                batch = self._apply_aug()(batch)
            elif not getattr(self, "_warned_no_aug", False):
                # the augmentation pipeline is a HIP kernel (gdlhip.augment): there is no host path to fall back to
                logger.warning("training batch is not on a GPU: the augmentation pipeline (segmentation_dofa.py:91-121) "
                               "is skipped -- this build augments in on_after_batch_transfer on the device")
                self._warned_no_aug = True
        return batch

    # ------------------------------------------------------------------ step tails
    def _predict(self, logits: Tensor) -> Tensor:
        """segmentation_dofa.py:278-281."""
        if self.num_classes == 1:
            return (logits.sigmoid().squeeze(1) > self.threshold).long()
        return gnn.predict_mask(logits)  # softmax(dim=1).argmax(dim=1), one kernel

    def _log_loss(self, name: str, loss: Tensor, bs: int) -> None:
        self.log(name, loss, batch_size=bs, prog_bar=True, logger=True, on_step=False, on_epoch=True, sync_dist=True,
                 rank_zero_only=True)

    def _log_test_metrics(self, y_hat: Tensor, y: Tensor, loss: Tensor, bs: int) -> None:
        """segmentation_dofa.py:313-338 (figure logging, MLflow artifacts: out of scope)."""
        metrics = self.iou_classwise_metric(y_hat, y)      # per-class IoU from the integer count kernel
        self.iou_classwise_metric.reset()
        metrics["test_loss"] = loss
        self.log_dict(metrics, batch_size=bs, prog_bar=False, logger=True, on_step=False, sync_dist=True,
                      rank_zero_only=True)

    # ------------------------------------------------------------------ captured steps
    def on_graph_replay(self, batch: dict[str, Any]) -> None:
        """Called by ``gdlhip.graphs.GraphedTrainStep`` after every hipGraph replay of the training step: ``training_step`` itself
        does not run on the host any more, so its host-side bookkeeping (the sample counter of segmentation_dofa.py:234) is
        advanced here.  Anything else ``training_step`` does must live on the device."""
        img = batch.get("image") if isinstance(batch, dict) else None
        if isinstance(img, Tensor):
            self.train_samples_count += int(img.shape[0])

    # ------------------------------------------------------------------ epoch ends
    def on_train_epoch_end(self) -> None:
        logger.info("Training epoch complete. Processed %d samples", self.train_samples_count)
        self.train_samples_count = 0

    def on_validation_epoch_end(self) -> None:
        logger.info("Validation epoch complete. Processed %d samples", self.val_samples_count)
        self.val_samples_count = 0

    def on_test_epoch_end(self) -> None:
        logger.info("Test epoch complete. Processed %d samples", self.test_samples_count)
        self.test_samples_count = 0
