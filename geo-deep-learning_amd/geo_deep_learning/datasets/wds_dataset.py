"""Per-sample processing of the sharded tile dataset (hot-path part of the reference's
datasets/wds_dataset.py: ``_load_normalization_stats`` :198-215, ``_process_sample`` :217-244, the three
``_prepare_*_output`` layouts :246-306, the metadata encoders :308-389).

The reference converts every tile to f32 in the dataloader workers, normalises it on the CPU and ships
**f32** over PCIe.  Here the worker keeps the tile in its stored dtype (uint8 for every sensor the
reference's stats files describe): ``process_sample(..., defer_normalization=True)`` returns the same batch
dict, with ``image`` raw and the sensor's ``mean`` / ``std`` beside it, and
``geo_deep_learning.datamodules.device_input.DeviceInputStage`` finishes ``x/255 -> (x-mean)/std`` in one
HBM-bound HIP kernel after an asynchronous pinned-memory H2D copy (4x fewer PCIe bytes for uint8).
``defer_normalization=False`` reproduces the reference's host arithmetic exactly; it is what
``MultiSensorDataModule`` selects when no ``device=`` is given, so the reference's data config yields the reference's
normalised f32 batches.

The shard side (``load_sensor_configs`` :46-49, ``create_shard_split_paths`` :52-79, ``create_sensor_datasets``
:82-137, ``ShardedDataset.build_web_dataset`` :391-431) is mirrored too, on a small built-in reader of the WebDataset
tar layout (``<key>.image_patch.npy`` / ``<key>.label_patch.npy`` / ``<key>.metadata.json`` members grouped by key) --
the third-party ``webdataset`` package is not required.
"""

from __future__ import annotations

import io
import json
import logging
import math
import random
import tarfile
from collections.abc import Iterator
from datetime import datetime
from pathlib import Path
from typing import Any

import numpy as np
import torch
import yaml

from geo_deep_learning.utils.tensors import normalization, standardization

logger = logging.getLogger(__name__)

DEFAULT_WAVELENGTH_KEYS = ["red_wavelength", "green_wavelength", "blue_wavelength", "nir_wavelength"]


def load_normalization_stats(stats_path: str, sensor_name: str) -> dict[str, Any]:
    """wds_dataset.py:198-215: per-sensor mean / std divided by 255, shaped [C,1,1]."""
    with Path(stats_path).open() as f:
        data = json.load(f)
    stats = data["statistics"][sensor_name]
    return {
        "mean": torch.tensor(stats["mean"], dtype=torch.float32).div(255.0).view(-1, 1, 1),
        "std": torch.tensor(stats["std"], dtype=torch.float32).div(255.0).view(-1, 1, 1),
        "band_count": stats["band_count"],
        "patch_count": stats["patch_count"],
        "dtype": stats["dtype"],
    }


def encode_temporal(datetime_str: str) -> torch.Tensor:
    """Week-of-year / hour-of-day sin-cos encoding (wds_dataset.py:308-340)."""
    try:
        if datetime_str.endswith("Z"):
            datetime_str = datetime_str[:-1] + "+00:00"
        dt = datetime.fromisoformat(datetime_str)
        week_rad = (dt.isocalendar().week / 52.0) * 2 * math.pi
        hour_rad = (dt.hour / 24.0) * 2 * math.pi
        return torch.tensor([math.sin(week_rad), math.cos(week_rad), math.sin(hour_rad), math.cos(hour_rad)],
                            dtype=torch.float32)
    except Exception as e:  # noqa: BLE001
        logger.warning("Error parsing datetime: %s %s", datetime_str, e)
        return torch.zeros(4, dtype=torch.float32)


def encode_spatial(lat: float, lon: float) -> torch.Tensor:
    """lat / lon sin-cos encoding (wds_dataset.py:342-362)."""
    try:
        la, lo = math.radians(lat), math.radians(lon)
        return torch.tensor([math.sin(la), math.cos(la), math.sin(lo), math.cos(lo)], dtype=torch.float32)
    except Exception as e:  # noqa: BLE001
        logger.warning("Error parsing coordinates: %s %s %s", lat, lon, e)
        return torch.zeros(4, dtype=torch.float32)


class SampleProcessor:
    """The arithmetic half of the reference's ``ShardedDataset`` (one instance per sensor / split)."""

    # integer tiles may stay raw for DeviceInputStage (which recognises exactly these dtypes); f32 tiles are always
    # normalised here, as the reference does (wds_dataset.py:230-236)
    RAW_DTYPES = (torch.uint8, torch.uint16, torch.int16)

    def __init__(self, sensor_name: str, norm_stats: dict[str, Any], model_type: str = "dofa",
                 wavelength_keys: list[str] | None = None, *, defer_normalization: bool = False) -> None:
        self.sensor_name = sensor_name
        self.norm_stats = norm_stats
        self.model_type = model_type
        self.wavelength_keys = wavelength_keys
        self.wavelengths_cache: dict[str, torch.Tensor] = {}
        self.defer_normalization = defer_normalization

    def __call__(self, sample: dict[str, Any]) -> dict[str, Any]:
        return self.process_sample(sample)

    def process_sample(self, sample: dict[str, Any]) -> dict[str, Any]:
        """wds_dataset.py:217-244.  ``sample`` = {"__key__", "image_patch.npy", "label_patch.npy",
        "metadata.json"}."""
        raw = torch.from_numpy(np.ascontiguousarray(sample["image_patch.npy"]))
        label = torch.from_numpy(sample["label_patch.npy"]).long()
        metadata = sample["metadata.json"]
        if self.defer_normalization and raw.dtype in self.RAW_DTYPES:
            image = raw                     # finished on the GPU by DeviceInputStage
        else:
            image = normalization(raw.float())
            image = standardization(image, self.norm_stats["mean"], self.norm_stats["std"])
        out = {"image": image, "mask": label, "platform": self.sensor_name, "image_name": sample["__key__"]}
        if self.model_type == "clay":        # :246-270
            meta = metadata["metadata"]
            out["time"] = encode_temporal(meta.get("datetime", "0.0"))
            out["latlon"] = encode_spatial(meta.get("coordinates_lat", 0.0), meta.get("coordinates_lon", 0.0))
        elif self.model_type == "dofa":      # :272-288
            out["wavelengths"] = self.extract_wavelengths(metadata)
        else:                                # :290-306
            out["metadata"] = metadata
        out["mean"] = self.norm_stats["mean"]
        out["std"] = self.norm_stats["std"]
        return out

    def extract_wavelengths(self, metadata: dict[str, Any]) -> torch.Tensor:
        """wds_dataset.py:364-389."""
        keys = self.wavelength_keys or DEFAULT_WAVELENGTH_KEYS
        try:
            meta = metadata["metadata"]
            wavelengths = [float(meta[band]) for band in keys if band in meta]
            cache_key = f"{self.sensor_name}_{'_'.join(keys)}"
            if cache_key not in self.wavelengths_cache:
                self.wavelengths_cache[cache_key] = torch.tensor(wavelengths, dtype=torch.float32)
            return self.wavelengths_cache[cache_key]
        except Exception as e:  # noqa: BLE001
            logger.warning("Error extracting wavelengths: %s", e)
            return torch.tensor([0.0] * len(keys), dtype=torch.float32)


def collate(samples: list[dict[str, Any]]) -> dict[str, Any]:
    """Batch a list of processed samples the way the reference's ``.batched(batch_size)`` + default collation
    does: tensors stacked, strings / dicts listed."""
    out: dict[str, Any] = {}
    for k in samples[0]:
        vals = [s[k] for s in samples]
        out[k] = torch.stack(vals) if isinstance(vals[0], torch.Tensor) else vals
    return out


# ------------------------------------------------------------------ shards
def load_sensor_configs(config_path: str) -> dict[str, dict[str, str]]:
    """wds_dataset.py:46-49."""
    with Path(config_path).open() as f:
        return yaml.safe_load(f)


def create_shard_split_paths(manifest_path: str, split: str, parent_dir: str | None = None) -> tuple[list[str], int]:
    """wds_dataset.py:52-79."""
    parent = Path(manifest_path).parent / split if parent_dir is None else Path(parent_dir) / split
    with Path(manifest_path).open() as f:
        data = json.load(f)
    return ([(parent / item["path"]).as_posix() for item in data["shards"][split]],
            data["statistics"]["patch_counts"][split])


def read_tar_samples(path: str) -> Iterator[dict[str, Any]]:
    """One WebDataset shard: members are grouped by the basename up to its first dot; ``.npy`` members are decoded
    to arrays, ``.json`` to objects (what ``wds.WebDataset(...).decode()`` yields for these extensions)."""
    cur: dict[str, Any] = {}
    with tarfile.open(path, "r") as tf:
        for member in tf:
            if not member.isfile():
                continue
            name = Path(member.name).name
            key, _, ext = name.partition(".")
            prefix = (Path(member.name).parent / key).as_posix().lstrip("./")
            if cur and cur["__key__"] != prefix:
                yield cur
                cur = {}
            cur.setdefault("__key__", prefix)
            data = tf.extractfile(member).read()
            if ext.endswith(".npy") or ext == "npy":
                cur[ext] = np.load(io.BytesIO(data), allow_pickle=False)
            elif ext.endswith(".json") or ext == "json":
                cur[ext] = json.loads(data)
            else:
                cur[ext] = data
    if cur:
        yield cur


class ShardedDataset:
    """wds_dataset.py:140-431: one sensor / split.  ``build_web_dataset()`` returns an iterable of BATCHES (dicts),
    like the reference's ``.decode().map(_process_sample).batched(batch_size, partial=split != "trn")``."""

    def __init__(self, sensor_name: str, shard_paths: list[str], patch_count: int, normalization_stats_path: str,
                 model_type: str = "clay", split: str = "trn", batch_size: int = 16, shuffle_buffer: int = 1000,
                 shardshuffle: int | None = None, seed: int = 42, epoch_size: int | None = None,
                 wavelength_keys: list[str] | None = None, *, defer_normalization: bool = False) -> None:
        self.sensor_name, self.shard_paths, self.patch_count = sensor_name, shard_paths, patch_count
        self.model_type, self.split, self.batch_size = model_type, split, batch_size
        self.shuffle_buffer, self.shardshuffle, self.seed, self.epoch_size = shuffle_buffer, shardshuffle, seed, epoch_size
        self.norm_stats = load_normalization_stats(normalization_stats_path, sensor_name)
        self.processor = SampleProcessor(sensor_name, self.norm_stats, model_type, wavelength_keys,
                                         defer_normalization=defer_normalization)

    def _process_sample(self, sample: dict[str, Any]) -> dict[str, Any]:
        return self.processor.process_sample(sample)

    def _shards(self) -> list[str]:
        # trn: sliced by rank (:398-401); val: wds.split_by_node does the same slice (:415); tst: every rank reads all
        # shards (nodesplitter=None, :415).  (The reference's trn pipeline slices a second time through split_by_node --
        # rank r then sees only every world-th shard of its own slice; that is not reproduced.)
        shard_list = sorted(self.shard_paths)
        if self.split in ("trn", "val") and torch.distributed.is_available() and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
            if self.split == "val" and len(shard_list) < world:
                # fewer validation shards than ranks: a slice would leave some ranks without a single batch (no metrics, no
                # monitored value for checkpointing / early stopping).  Every rank validates on all shards instead -- the
                # rank-averaged epoch means are then the single-process means
                return shard_list
            shard_list = shard_list[rank::world]
        return shard_list

    def _samples(self, epoch: int) -> Iterator[dict[str, Any]]:
        shards = self._shards()
        rng = random.Random((self.seed or 0) + epoch)
        if self.split == "trn" and self.shardshuffle:
            rng.shuffle(shards)
        buf: list = []
        for path in shards:
            for sample in read_tar_samples(path):
                if self.split != "trn" or not self.shuffle_buffer:
                    yield sample
                    continue
                buf.append(sample)                      # wds .shuffle(n): reservoir of n samples
                if len(buf) >= self.shuffle_buffer:
                    yield buf.pop(rng.randrange(len(buf)))
        while buf:
            yield buf.pop(rng.randrange(len(buf)))

    def build_web_dataset(self) -> "_BatchPipeline":
        return _BatchPipeline(self)


class _BatchPipeline:
    def __init__(self, ds: ShardedDataset) -> None:
        self.ds, self.epoch = ds, 0

    def __iter__(self) -> Iterator[dict[str, Any]]:
        ds = self.ds
        batch: list = []
        # the epoch counter advances when an iteration STARTS (like wds' shard lists do): a consumer that stops early
        # (``epoch_size``) still gets a different shard / shuffle order next time
        epoch, self.epoch = self.epoch, self.epoch + 1
        for sample in ds._samples(epoch):
            try:
                batch.append(ds._process_sample(sample))
            except Exception as e:  # noqa: BLE001  (wds.warn_and_continue)
                logger.warning("skipping sample %s: %s", sample.get("__key__"), e)
                continue
            if len(batch) == ds.batch_size:
                yield collate(batch)
                batch = []
        if batch and ds.split != "trn":
            yield collate(batch)                        # partial batches only outside training (:430)


def create_sensor_datasets(sensor_configs_path: str, **common_kwargs: object) -> dict[str, Any]:
    """wds_dataset.py:82-137."""
    datasets: dict[str, Any] = {}
    for sensor_name, config in load_sensor_configs(sensor_configs_path).items():
        datasets[sensor_name] = {}
        for split in ("trn", "val", "tst"):
            try:
                shard_paths, patch_count = create_shard_split_paths(config["manifest_path"], split, config["parent_dir"])
            except Exception:
                logger.exception("Failed to create dataset for %s %s split", sensor_name, split)
                continue
            if not shard_paths:
                logger.warning("No shards found for %s %s split", sensor_name, split)
                continue
            datasets[sensor_name][split] = ShardedDataset(
                sensor_name=sensor_name, shard_paths=shard_paths, patch_count=patch_count,
                normalization_stats_path=config["stats_path"], split=split,
                wavelength_keys=config.get("wavelength_keys"), **common_kwargs)
    return datasets
