#!/usr/bin/env python
"""Tuning probe (round 6): the forward gather-sum of UperNet's fpn_bottleneck (DOFA-base: 256 output channels, sources 72^2 / 36^2 /
18^2 -> 144^2) at per-GPU batch B: version 3 (three rolling rings) against version 1.  usage: bench_tapsum_roll3.py [B]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

lib = _lib.load()
lib.gdl_debug_set_tapsum_roll.argtypes = [ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 256
zs = [torch.randn(B, 144 // f, 144 // f, 9 * N, device="cuda").to(torch.bfloat16) for f in (2, 4, 8)]
gb = (sum(z.numel() for z in zs) + B * 144 * 144 * N) * 2 / 1e9


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for roll in (1, 0, 1, 0):
    lib.gdl_debug_set_tapsum_roll(roll)
    t = timed(lambda: ops.resize_conv3x3_fwd_sum(zs, (144, 144)))
    print(f"fpn_bottleneck sources B={B} roll={roll}: {t:7.1f} us = {gb / t * 1e3:5.2f} TB/s (algorithmic {gb:.2f} GB)", flush=True)
lib.gdl_debug_set_tapsum_roll(1)
