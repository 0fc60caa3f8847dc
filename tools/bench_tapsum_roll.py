#!/usr/bin/env python
"""Tuning probe (round 6): the forward gather-sum of the neck's x4 and x2 levels (DOFA-base, 768 channels, 36^2 -> 144^2 / 72^2) at
per-GPU batch B, plain and with BatchNorm statistics: version 3 (rolling row window) against versions 1 / 2.
usage: bench_tapsum_roll.py [B]"""
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
N = 768
add = torch.randn(N, device="cuda")


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


for f in (4, 2):
    z = torch.randn(B, 36, 36, 9 * N, device="cuda").to(torch.bfloat16)
    size = (36 * f, 36 * f)
    gb = (z.numel() + B * size[0] * size[1] * N) * 2 / 1e9
    for roll in (1, 0, 1, 0):
        lib.gdl_debug_set_tapsum_roll(roll)
        t0 = timed(lambda: ops.resize_conv3x3_fwd_sum([z], size, addvec=add))
        t1 = timed(lambda: ops.resize_conv3x3_fwd_sum_bn([z], size, addvec=add))
        print(f"x{f} B={B} roll={roll}: plain {t0:7.1f} us = {gb / t0 * 1e3:5.2f} TB/s   with statistics {t1:7.1f} us = {gb / t1 * 1e3:5.2f} TB/s "
              f"(algorithmic {gb:.2f} GB)", flush=True)
lib.gdl_debug_set_tapsum_roll(1)
