"""Per-tile preprocess (drop-in for the reference's utils/tensors.py:10-35).

``normalization`` / ``standardization`` keep the reference's signatures for host tensors (they
are plain elementwise formulas used by the dataset workers); on the GPU path the fused
``normalize_standardize_u8`` ships uint8 tiles over PCIe (4x fewer bytes than the reference's
f32, datasets/wds_dataset.py:230-236) and applies both steps in ONE HBM-bound HIP kernel."""

import torch

from gdlhip import ops


def normalization(input_tensor, image_min=0, image_max=255, norm_min=0.0, norm_max=1.0):
    shape = input_tensor.shape
    out = (norm_max - norm_min) * (input_tensor - image_min) / (image_max - image_min) + norm_min
    return out.reshape(shape)


def standardization(input_tensor, mean, std):
    shape = input_tensor.shape
    b, c = input_tensor.shape[:2]
    x = input_tensor.reshape(b, c, -1)
    return ((x - mean) / std).reshape(shape)


def normalize_standardize_u8(tile_u8: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """uint8 [B,C,H,W] on the GPU -> standardised f32, == standardization(normalization(x.float()))."""
    return ops.normalize_u8(tile_u8, mean.reshape(-1).float().contiguous(), std.reshape(-1).float().contiguous())
