"""A CPU-only task + datamodule with the hook surface of the reference's tasks (for MiniTrainer / train.py tests)."""

from __future__ import annotations

import torch
from torch import nn

from gdlhip.trainer import LightningModule


class ToyNet(nn.Module):
    def __init__(self, num_classes: int, batchnorm: bool = True) -> None:
        super().__init__()
        self.encoder = nn.Conv2d(3, 8, 3, padding=1)
        self.norm = nn.BatchNorm2d(8) if batchnorm else nn.Identity()
        self.head = nn.Conv2d(8, num_classes, 1)

    def forward(self, x):
        return self.head(torch.relu(self.norm(self.encoder(x))))


class ToyTask(LightningModule):
    def __init__(self, num_classes: int, loss, optimizer=torch.optim.Adam, scheduler=torch.optim.lr_scheduler.ConstantLR,
                 scheduler_config=None, **kwargs) -> None:
        super().__init__()
        self.save_hyperparameters(num_classes=num_classes, **kwargs)
        self.num_classes, self.loss = num_classes, loss
        self.optimizer, self.scheduler = optimizer, scheduler
        self.scheduler_config = scheduler_config or {"interval": "epoch"}
        self.calls: list[str] = []

    def configure_model(self):
        if getattr(self, "model", None) is None:
            self.model = ToyNet(self.num_classes, self.hparams.get("batchnorm", True))

    def configure_optimizers(self):
        opt = self.optimizer(self.parameters())
        return [opt], [{"scheduler": self.scheduler(opt), **self.scheduler_config}]

    def forward(self, x):
        return self.model(x)

    def on_before_batch_transfer(self, batch, dataloader_idx):
        self.calls.append("before")
        return batch

    def on_after_batch_transfer(self, batch, dataloader_idx):
        self.calls.append("after:train" if self.trainer.training else "after:eval")
        return batch

    def _loss(self, batch):
        return self.loss(self(batch["image"]), batch["mask"].squeeze(1).long())

    def training_step(self, batch, batch_idx):
        loss = self._loss(batch)
        self.log("train_loss", loss, batch_size=batch["image"].shape[0])
        return loss

    def validation_step(self, batch, batch_idx):
        self.log("val_loss", self._loss(batch), batch_size=batch["image"].shape[0])

    def test_step(self, batch, batch_idx):
        self.log_dict({"test_loss": self._loss(batch)}, batch_size=batch["image"].shape[0])

    def on_train_epoch_end(self):
        self.calls.append("train_epoch_end")

    def on_validation_epoch_end(self):
        self.calls.append("val_epoch_end")

    def on_test_epoch_end(self):
        self.calls.append("test_epoch_end")


def make_batches(n: int, batch_size: int, num_classes: int, seed: int, rank: int = 0, world: int = 1):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        img = torch.randn(batch_size, 3, 8, 8, generator=g)
        mask = (img[:, :1] * 1.5 + 1.5).clamp(0, num_classes - 1).long()   # learnable from the image
        out.append({"image": img[rank::world], "mask": mask[rank::world]})
    return out


class ToyData:
    """Datamodule surface: setup / train_dataloader / val_dataloader / test_dataloader."""

    def __init__(self, batch_size: int = 4, num_classes: int = 3, train_batches: int = 6, with_test: bool = True) -> None:
        self.batch_size, self.num_classes, self.train_batches, self.with_test = batch_size, num_classes, train_batches, with_test
        self.epoch_size = None

    def setup(self, stage=None):
        self.trn = make_batches(self.train_batches, self.batch_size, self.num_classes, 1)
        self.val = make_batches(2, self.batch_size, self.num_classes, 2)
        self.tst = make_batches(2, self.batch_size, self.num_classes, 3)

    def train_dataloader(self):
        return self.trn

    def val_dataloader(self):
        return self.val

    def test_dataloader(self):
        return self.tst if self.with_test else None
