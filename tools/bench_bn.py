#!/usr/bin/env python
"""GPU micro-benchmark of the BatchNorm kernels on the decoder's large maps (batch 32, bf16): GB/s of algorithmic traffic
(statistics: read x; apply: read x, write y; backward sums: read x and dy; backward dx: read x and dy, write dx).
bn_bwd_reduce is timed with the 16-byte-load kernel and with the round-2 mapping (gdl_debug_set_bn_wide)."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

lib = _lib.load()
lib.gdl_debug_set_bn_wide.argtypes = [ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bf = torch.bfloat16


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
    return ts[len(ts) // 2] * 1e3


for name, hw, c in (("neck x4", 144, 768), ("neck x2", 72, 768), ("fpn 144", 144, 256), ("fpn 72", 72, 256), ("neck x1", 36, 768), ("unet++ 256", 256, 64), ("unet++ 512", 512, 16)):
    x = torch.randn(B, hw, hw, c, device="cuda").to(bf)
    dy = torch.randn(B, hw, hw, c, device="cuda").to(bf)
    g, b = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    mean, var = ops.bn_stats(x, rm, rv, 0.1)
    nbytes = x.numel() * 2
    t_stats = timeit(lambda: ops.bn_stats(x, rm, rv, 0.1))
    t_apply = timeit(lambda: ops.bn_apply(x, mean, var, g, b, 1e-5, True))
    res = {}
    for wide in (1, 0):
        lib.gdl_debug_set_bn_wide(wide)
        res[wide] = (timeit(lambda: ops.bn_bwd_reduce(x, dy, mean, var, g, b, 1e-5, True)), ops.bn_bwd_reduce(x, dy, mean, var, g, b, 1e-5, True))
    lib.gdl_debug_set_bn_wide(1)
    dg, db = res[1][1]
    dev = max(((res[1][1][i] - res[0][1][i]).abs().max() / res[0][1][i].abs().max()).item() for i in (0, 1))
    t_dx = timeit(lambda: ops.bn_bwd_dx(x, dy, mean, var, g, b, 1e-5, True, dg, db, B * hw * hw))
    print(f"{name:8s} [{B},{hw},{hw},{c}] {nbytes / 1e6:6.0f} MB: stats {t_stats:6.0f} us ({nbytes / t_stats / 1e3:5.0f} GB/s) | apply {t_apply:6.0f} us "
          f"({2 * nbytes / t_apply / 1e3:5.0f} GB/s) | bwd sums 16-byte {res[1][0]:6.0f} us ({2 * nbytes / res[1][0] / 1e3:5.0f} GB/s) vs round 2 "
          f"{res[0][0]:6.0f} us ({2 * nbytes / res[0][0] / 1e3:5.0f} GB/s), rel diff {dev:.1e} | bwd dx {t_dx:6.0f} us ({3 * nbytes / t_dx / 1e3:5.0f} GB/s)", flush=True)
