#!/usr/bin/env python
"""GPU micro-benchmark of conv_gemm tile variants on narrow-output (N <= 64) layers of UNet++ / ResNet / MiT stage 1
at per-GPU batch 32: 64^2 tile (variant 0) vs 128^2 (1) vs 256x64 (5), interleaved in one process."""

import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DEV, bf = "cuda", torch.bfloat16
LAYERS = [  # name, H, W, C, N, R
    ("unet++ 3x3 64->64 @256", 256, 256, 64, 64, 3),
    ("unet++ 3x3 128->64 @256", 256, 256, 128, 64, 3),
    ("unet++ 3x3 64->64 @512", 512, 512, 64, 64, 3),
    ("resnet 3x3 64->64 @128", 128, 128, 64, 64, 3),
    ("mit fc2 256->64 @128 (1x1)", 128, 128, 256, 64, 1),
    ("mit q 64->64 @128 (1x1)", 128, 128, 64, 64, 1),
]


def timeit(fn, rounds=5, inner=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    print(f"batch {B}: TF/s (us)  auto | 64^2 | 128^2 | 256x64")
    for name, h, w, c, n, r in LAYERS:
        x = torch.randn(B, h, w, c, device=DEV).to(bf)
        wt = (torch.randn(n, r * r * c, device=DEV) * 0.05).to(bf)
        out = torch.empty(B, h, w, n, device=DEV, dtype=bf)
        flops = 2 * B * h * w * n * r * r * c
        res = []
        for v in (-1, 0, 1, 5):
            lib.gdl_debug_force_conv_variant(v)
            try:
                t = timeit(lambda: ops.conv_gemm(x, wt, R=r, S=r, pad=r // 2, out=out))
            finally:
                lib.gdl_debug_force_conv_variant(-1)
            res.append(f"{flops / t / 1e9:7.1f} ({t * 1e3:6.0f})")
        print(f"{name:30s} GF {flops / 1e9:8.1f}  " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
