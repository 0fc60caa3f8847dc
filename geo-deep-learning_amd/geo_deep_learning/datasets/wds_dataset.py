"""Per-sample processing of the sharded tile dataset (hot-path part of the reference's
datasets/wds_dataset.py: ``_load_normalization_stats`` :198-215, ``_process_sample`` :217-244, the three
``_prepare_*_output`` layouts :246-306, the metadata encoders :308-389).

The reference converts every tile to f32 in the dataloader workers, normalises it on the CPU and ships
**f32** over PCIe.  Here the worker keeps the tile in its stored dtype (uint8 for every sensor the
reference's stats files describe): ``process_sample(..., defer_normalization=True)`` returns the same batch
dict, with ``image`` raw and the sensor's ``mean`` / ``std`` beside it, and
``geo_deep_learning.datamodules.device_input.DeviceInputStage`` finishes ``x/255 -> (x-mean)/std`` in one
HBM-bound HIP kernel after an asynchronous pinned-memory H2D copy (4x fewer PCIe bytes for uint8).
``defer_normalization=False`` reproduces the reference's host arithmetic exactly (used by the parity tests).

Reading the WebDataset tar shards themselves (third-party ``webdataset``; I/O, not arithmetic) is out of scope:
any iterable of sample dicts with the reference's keys can be fed to :class:`SampleProcessor`.
"""

from __future__ import annotations

import json
import logging
import math
from datetime import datetime
from pathlib import Path
from typing import Any

import numpy as np
import torch

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

    RAW_DTYPES = (torch.uint8, torch.uint16, torch.int16, torch.float32)

    def __init__(self, sensor_name: str, norm_stats: dict[str, Any], model_type: str = "dofa",
                 wavelength_keys: list[str] | None = None, *, defer_normalization: bool = True) -> None:
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
