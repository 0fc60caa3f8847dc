#!/usr/bin/env python
"""GPU micro-benchmark of the bf16 weight-gradient kernels on 3x3 layers of the three models (per-GPU batch 32 shapes):
auto selection (row-segment kernel wherever the width allows) vs mode 8 (N, C >= 256 layers on the 256^2 per-tap
kernel) and mode 3 (per-tap 128^2 kernel), interleaved in ONE
process so clocks / box variance cancel.

    python tools/bench_wgrad.py [batch]
"""

import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DEV = "cuda"
bf = torch.bfloat16

LAYERS = [  # name, H, W, C, N
    ("upernet fpn 256->256 @128", 128, 128, 256, 256),
    ("upernet fpn 256->256 @64", 64, 64, 256, 256),
    ("upernet fuse 1024->256 @128", 128, 128, 1024, 256),
    ("upernet psp 2816->256 @16", 16, 16, 2816, 256),
    ("neck 768->768 @128", 128, 128, 768, 768),
    ("neck 768->768 @64", 64, 64, 768, 768),
    ("dofa@512 neck 768->768 @144", 144, 144, 768, 768),
    ("dofa@512 neck 768->768 @72", 72, 72, 768, 768),
    ("dofa@512 fuse 1024->256 @144", 144, 144, 1024, 256),
    ("dofa@512 fpn 256->256 @144", 144, 144, 256, 256),
    ("dofa@512 fpn 256->256 @36", 36, 36, 256, 256),
    ("unet++ 64->64 @256", 256, 256, 64, 64),
    ("unet++ 128->64 @256", 256, 256, 128, 64),
    ("unet++ 192->64 @128", 128, 128, 192, 64),
    ("unet++ 128->128 @64", 64, 64, 128, 128),
    ("unet++ 64(16)->64(16) @512", 512, 512, 64, 64),
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
    lib.gdl_debug_force_wgrad_small.argtypes = [ctypes.c_int]
    print(f"batch {B}: TF/s (us)  auto | wide layers on the 256^2 per-tap kernel | per-tap 128^2")
    for name, h, w, c, n in LAYERS:
        b = B if h * w * max(c, n) * B * 2 < (1 << 31) else B // 2
        x = torch.randn(b, h, w, c, device=DEV).to(bf)
        dy = torch.randn(b, h, w, n, device=DEV).to(bf)
        flops = 2 * b * h * w * n * 9 * c
        out = []
        ref = None
        for mode in (0, 8, 3):
            lib.gdl_debug_force_wgrad_small(mode)
            try:
                t = timeit(lambda: ops.conv_wgrad(x, dy, R=3, S=3, pad=1))
                got = ops.conv_wgrad(x, dy, R=3, S=3, pad=1)
            finally:
                lib.gdl_debug_force_wgrad_small(0)
            if ref is None:
                ref = got
            err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
            out.append(f"{flops / t / 1e9:7.1f} ({t * 1e3:6.0f}) e={err:.0e}")
        print(f"{name:30s} b={b:2d} GF {flops / 1e9:8.1f}  " + " | ".join(out), flush=True)


SPLIT_LAYERS = [  # name, H, W, C, N, R: the layers of the DOFA training step that run on the 256^2 kernel
    ("neck lateral 768->768 1x1 @36", 36, 36, 768, 768, 1), ("neck taps 768->6912 1x1 @36", 36, 36, 768, 6912, 1),
    ("fuse taps 256->2304 1x1 @72", 72, 72, 256, 2304, 1), ("fuse taps 256->2304 1x1 @36", 36, 36, 256, 2304, 1),
    ("fuse taps 256->2304 1x1 @18", 18, 18, 256, 2304, 1), ("lateral 768->256 1x1 @144", 144, 144, 768, 256, 1),
    ("lateral 768->256 1x1 @72", 72, 72, 768, 256, 1), ("lateral 768->256 1x1 @36", 36, 36, 768, 256, 1),
    ("neck 768->768 3x3 @36 (rows kernel)", 36, 36, 768, 768, 3), ("neck 768->768 3x3 @18", 18, 18, 768, 768, 3),
    ("psp bottleneck 1792->256 3x3 @18", 18, 18, 1792, 256, 3), ("fpn 768->256 3x3 @18", 18, 18, 768, 256, 3),
    ("fpn 256->256 3x3 @36 (rows kernel)", 36, 36, 256, 256, 3),
]


def main_splits():
    """tools/bench_wgrad.py <batch> splits: split-K rule of the 256^2 kernel, round 3 ("about 512 workgroups") vs the round-4 cost
    model, and for the 3x3 layers also the per-tap 256^2 kernel instead of the row-segment kernel (mode 8)."""
    lib = _lib.load()
    lib.gdl_debug_force_wgrad_small.argtypes = [ctypes.c_int]
    lib.gdl_debug_set_wgrad_old_splits.argtypes = [ctypes.c_int]
    print(f"batch {B}: TF/s (us)  round-3 splits | cost model | cost model, 3x3 on the per-tap 256^2 kernel")
    tot = [0.0, 0.0, 0.0]
    for name, h, w, c, n, r in SPLIT_LAYERS:
        x = torch.randn(B, h, w, c, device=DEV).to(bf)
        dy = torch.randn(B, h, w, n, device=DEV).to(bf)
        flops = 2 * B * h * w * n * r * r * c
        out, ref = [], None
        for i, (old, mode) in enumerate(((1, 0), (0, 0), (0, 8))):
            lib.gdl_debug_set_wgrad_old_splits(old)
            lib.gdl_debug_force_wgrad_small(mode)
            try:
                t = timeit(lambda: ops.conv_wgrad(x, dy, R=r, S=r, pad=r // 2))
                got = ops.conv_wgrad(x, dy, R=r, S=r, pad=r // 2)
            finally:
                lib.gdl_debug_set_wgrad_old_splits(0)
                lib.gdl_debug_force_wgrad_small(0)
            ref = got if ref is None else ref
            err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
            tot[i] += t * 1e3
            out.append(f"{flops / t / 1e9:7.1f} ({t * 1e3:6.0f}) e={err:.0e}")
        print(f"{name:38s} GF {flops / 1e9:7.1f}  " + " | ".join(out), flush=True)
    print("sum: " + " | ".join(f"{t:.0f} us" for t in tot))


def main_xcd():
    """tools/bench_wgrad.py <batch> xcd: the row-segment kernel with its tiles in launch order (round 4) vs all tiles of a pixel range
    on one XCD (round 5), interleaved, every 3x3 layer of LAYERS that takes that kernel."""
    lib = _lib.load()
    lib.gdl_debug_set_wgrad_rows_xcd.argtypes = [ctypes.c_int]
    print(f"batch {B}: TF/s (us)  launch order | XCD-grouped | launch order again | XCD-grouped again")
    for name, h, w, c, n in LAYERS:
        b = B if h * w * max(c, n) * B * 2 < (1 << 31) else B // 2
        x = torch.randn(b, h, w, c, device=DEV).to(bf)
        dy = torch.randn(b, h, w, n, device=DEV).to(bf)
        flops = 2 * b * h * w * n * 9 * c
        out, ref = [], None
        for on in (0, 1, 0, 1):
            lib.gdl_debug_set_wgrad_rows_xcd(on)
            try:
                t = timeit(lambda: ops.conv_wgrad(x, dy, R=3, S=3, pad=1))
                got = ops.conv_wgrad(x, dy, R=3, S=3, pad=1)
            finally:
                lib.gdl_debug_set_wgrad_rows_xcd(1)      # (the default)
            ref = got if ref is None else ref
            out.append(f"{flops / t / 1e9:7.1f} ({t * 1e3:6.0f}) ==:{torch.equal(got, ref)}")
        print(f"{name:30s} b={b:2d} GF {flops / 1e9:8.1f}  " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "xcd":
        main_xcd()
    elif len(sys.argv) > 2 and sys.argv[2] == "splits":
        main_splits()
    else:
        main()
