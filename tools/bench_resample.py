#!/usr/bin/env python
"""GPU micro-benchmark of the bf16 bilinear resamples of DOFA-base + UperNet at batch 32 (neck x4 / x2 / x0.5, decoder
upsamples to 144^2): row-structured kernels vs the flat-index ones, interleaved in one process; algorithmic GB/s."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

lib = _lib.load()
lib.gdl_debug_set_flat_resample.argtypes = [ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bf = torch.bfloat16


def timeit(fn, rounds=5, inner=4):
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


for name, hi, ho, c in (("neck 36->144 x768", 36, 144, 768), ("neck 36->72 x768", 36, 72, 768), ("neck 36->18 x768", 36, 18, 768),
                        ("decoder 72->144 x256", 72, 144, 256), ("decoder 36->144 x256", 36, 144, 256), ("decoder 18->144 x256", 18, 144, 256)):
    x = torch.randn(B, hi, hi, c, device="cuda").to(bf)
    y = torch.empty(B, ho, ho, c, device="cuda", dtype=bf)
    nbytes = (x.numel() + y.numel()) * 2
    out = []
    for flat in (0, 1):
        lib.gdl_debug_set_flat_resample(flat)
        tf = timeit(lambda: ops.bilinear(x, (ho, ho), out=y))
        tb = timeit(lambda: ops.bilinear_bwd(y, (hi, hi)))
        out.append(f"fwd {tf:6.0f} us {nbytes / tf / 1e3:6.0f} GB/s, bwd {tb:6.0f} us {nbytes / tb / 1e3:6.0f} GB/s")
    lib.gdl_debug_set_flat_resample(0)
    a = ops.bilinear(x, (ho, ho))
    g = ops.bilinear_bwd(a, (hi, hi))
    lib.gdl_debug_set_flat_resample(1)
    same = torch.equal(a, ops.bilinear(x, (ho, ho))) and torch.equal(g, ops.bilinear_bwd(a, (hi, hi)))
    lib.gdl_debug_set_flat_resample(0)
    # separable backward: vertical pass to [B, hi, ho, C], then horizontal pass (bilinear interpolation is a product of two
    # 1-D interpolations, so is its transpose); bf16 intermediate
    def two_pass():
        t = ops.bilinear_bwd(y, (hi, ho))
        return ops.bilinear_bwd(t, (hi, hi))
    t2 = timeit(two_pass) if ho > hi else float("nan")
    err = (two_pass().float() - g.float()).abs().max().item() / g.float().abs().max().item() if ho > hi else 0.0
    print(f"{name:24s} rows: {out[0]} | flat: {out[1]} | bit-identical {same} | two-pass bwd {t2:6.0f} us (rel dev {err:.1e})", flush=True)
