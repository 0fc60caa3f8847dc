"""Inference wrappers (drop-in for the reference's tools/script_model.py:10-86): raw 0-255 tile in, class
probabilities out.  The reference traces the model with ``torch.jit``; the HIP model is a sequence of C-ABI kernel
launches and needs no tracing, so the wrapper only fixes the pre/post-processing: ``/255 -> (x-mean)/std`` in the
fused normalise kernel, the model under ``no_grad`` (bf16 autocast optional), softmax / sigmoid in one kernel."""

from __future__ import annotations

import torch
from torch import nn

from gdlhip import ops


class ScriptModel(nn.Module):
    """tools/script_model.py:10-60."""

    def __init__(self, model: nn.Module, device: torch.device | None = None, num_classes: int = 1,
                 input_shape: tuple[int, int, int, int] = (1, 3, 512, 512), mean: list[float] | None = None,
                 std: list[float] | None = None, image_min: int = 0, image_max: int = 255, norm_min: float = 0.0,
                 norm_max: float = 1.0, *, from_logits: bool = True, bf16: bool = False) -> None:
        super().__init__()
        if (int(image_min), int(image_max), float(norm_min), float(norm_max)) != (0, 255, 0.0, 1.0):
            msg = "gdlhip ScriptModel: the fused normalise kernel implements the reference defaults (0..255 -> 0..1)"
            raise NotImplementedError(msg)
        self.device = device or torch.device("cuda")
        self.num_classes, self.from_logits, self.bf16 = num_classes, from_logits, bf16
        c = input_shape[1]
        self.register_buffer("mean", torch.tensor(mean or [0.0] * c, dtype=torch.float32))
        self.register_buffer("std", torch.tensor(std or [1.0] * c, dtype=torch.float32))
        self.model = model.eval()
        self.to(self.device)

    def _logits(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(self.device)
        if x.dtype not in (torch.uint8, torch.uint16, torch.int16):
            x = x.float()
        x = ops.normalize_raw(x.contiguous(), self.mean, self.std)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
            out = self._logits(x)
        if self.from_logits:
            out = ops.class_probs(out.float().contiguous())
        return out


class SegmentationScriptModel(ScriptModel):
    """tools/script_model.py:63-86: the model returns (out, aux); ``wavelengths`` are fixed at export time for DOFA."""

    def __init__(self, model: nn.Module, wavelengths: torch.Tensor | None = None, **kwargs: object) -> None:
        super().__init__(model, **kwargs)
        self.wavelengths = wavelengths

    def _logits(self, x: torch.Tensor) -> torch.Tensor:
        out = self.model(x) if self.wavelengths is None else self.model(x, self.wavelengths)
        return out[0] if isinstance(out, tuple) else out
