"""Multi-sensor data module (drop-in for the reference's datamodules/wds_datamodule.py:14-258: same constructor
keywords, ``setup`` / ``*_dataloader`` / ``teardown`` hooks, per-sensor batches mixed at random).

Differences that matter for the MI355X path: the loaders are thin prefetching iterables over the built-in tar reader
(datasets/wds_dataset.py) instead of ``webdataset.WebLoader`` worker processes, tiles stay in their stored dtype, and
``device=`` wraps each loader in :class:`DeviceInputStage` so the batches arrive on the GPU, normalised (and, for
training, augmented) by one kernel, two copies ahead of the step that consumes them."""

from __future__ import annotations

import logging
import queue
import random
import threading
from collections.abc import Iterator
from typing import Any

from geo_deep_learning.datasets.wds_dataset import create_sensor_datasets

try:  # Lightning is optional in this image
    from lightning.pytorch import LightningDataModule
except ImportError:  # pragma: no cover
    class LightningDataModule:  # type: ignore[no-redef]
        """Minimal stand-in exposing the hooks the reference's datamodule defines."""

        def __init__(self) -> None:
            pass

logger = logging.getLogger(__name__)


class RandomMix:
    """webdataset.RandomMix(datasets, probs=None, longest=True): every step draws one of the sources that still
    have batches, uniformly (wds_datamodule.py:231-243)."""

    def __init__(self, datasets: list, seed: int | None = None) -> None:
        self.datasets, self.seed, self.epoch = datasets, seed, 0

    def __iter__(self) -> Iterator[dict[str, Any]]:
        epoch, self.epoch = self.epoch, self.epoch + 1     # advances at the START of an iteration (early stops count)
        rng = random.Random(None if self.seed is None else self.seed + epoch)
        sources = [iter(d) for d in self.datasets]
        while sources:
            i = rng.randrange(len(sources))
            try:
                yield next(sources[i])
            except StopIteration:
                del sources[i]


class PrefetchLoader:
    """Decode batches on a background thread, ``prefetch`` ahead of the consumer (tile decode overlaps compute).
    ``epoch_batches`` reproduces ``WebLoader(batch_size=None).with_epoch(n)`` (wds_datamodule.py:104-113): an epoch is
    exactly n items of the loader -- BATCHES, because the dataset is already ``.batched()`` -- and a source that runs out
    earlier is started again (each restart is a new shard / shuffle epoch of the pipeline)."""

    def __init__(self, dataset: Any, prefetch: int = 2, epoch_batches: int | None = None) -> None:
        self.dataset, self.prefetch, self.epoch_batches = dataset, max(1, prefetch), epoch_batches

    def __iter__(self) -> Iterator[dict[str, Any]]:
        q: queue.Queue = queue.Queue(self.prefetch)
        stop = threading.Event()
        done = object()

        def work() -> None:
            try:
                n = 0
                while not stop.is_set():
                    got = 0
                    for b in self.dataset:
                        if stop.is_set() or (self.epoch_batches is not None and n >= self.epoch_batches):
                            break
                        q.put(b)
                        n += 1
                        got += 1
                    if self.epoch_batches is None or n >= self.epoch_batches or got == 0:
                        break
            except Exception as e:  # noqa: BLE001
                q.put(e)
            q.put(done)

        t = threading.Thread(target=work, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is done:
                    return
                if isinstance(item, Exception):
                    raise item
                yield item
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(0.01)


class MultiSensorDataModule(LightningDataModule):
    """wds_datamodule.py:14-258."""

    def __init__(self, sensor_configs_path: str, model_type: str = "clay", patch_size: tuple[int, int] = (512, 512),
                 epoch_size: int | None = None, batch_size: int = 16, num_workers: int = 0,
                 prefetch_factor: int | None = None, shuffle_buffer: int = 0, shardshuffle: int | None = None,
                 seed: int | None = None, *, device: str | None = None, augment: Any | None = None) -> None:
        super().__init__()
        self.sensor_configs_path, self.model_type, self.batch_size = sensor_configs_path, model_type, batch_size
        self.num_workers, self.prefetch_factor = num_workers, prefetch_factor
        self.shuffle_buffer, self.shardshuffle, self.seed = shuffle_buffer, shardshuffle, seed
        self.patch_size, self.epoch_size = patch_size, epoch_size
        self.device, self.augment = device, augment
        self.datasets: dict = {}
        self.train_loader = self.val_loader = self.test_loader = None

    def prepare_data(self) -> None:
        """Nothing to download."""

    def setup(self, stage: str | None = None) -> None:  # noqa: ARG002
        # raw (uint8) tiles + GPU normalisation only when a DeviceInputStage will finish them; with the reference's
        # data config (no ``device``) the batches are normalised on the host exactly like wds_dataset.py:217-244
        self.datasets = create_sensor_datasets(
            sensor_configs_path=self.sensor_configs_path, model_type=self.model_type, batch_size=self.batch_size,
            epoch_size=self.epoch_size, shuffle_buffer=self.shuffle_buffer, shardshuffle=self.shardshuffle,
            seed=self.seed if self.seed is not None else 42, defer_normalization=self.device is not None)
        self.train_loader = self._loader("trn")
        self.val_loader = self._loader("val")
        self.test_loader = self._loader("tst")

    def _loader(self, split: str):
        per_sensor = {n: s[split] for n, s in self.datasets.items() if split in s}
        if not per_sensor:
            (logger.info if split == "tst" else logger.warning)("No %s datasets found", split)
            return None
        pipes = [d.build_web_dataset() for d in per_sensor.values()]
        source = pipes[0] if len(pipes) == 1 else RandomMix(pipes, self.seed)
        epoch_batches = self.epoch_size if split == "trn" and self.epoch_size else None   # with_epoch counts batches
        loader: Any = PrefetchLoader(source, self.prefetch_factor or 2, epoch_batches)
        if self.device is not None:
            from geo_deep_learning.datamodules.device_input import DeviceInputStage
            loader = DeviceInputStage(loader, self.device, depth=2, augment=self.augment if split == "trn" else None)
        return loader

    def train_dataloader(self):
        return self.train_loader

    def val_dataloader(self):
        return self.val_loader

    def test_dataloader(self):
        return self.test_loader

    def teardown(self, stage: str | None = None) -> None:  # noqa: ARG002
        self.datasets.clear()
