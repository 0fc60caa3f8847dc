#!/usr/bin/env python
"""The 256^2 one-wave-per-SIMD tile (variant 8) against the 8-wave ping-pong tile (3) and the 3x3 shared-staging kernel (4) on the
deep-K layers of the DOFA + UperNet step (batch 32, bf16): 3x3 convolutions, their data gradients, the K >= 2304 GEMMs.
   tools/bench_w4.py [batch]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bf = torch.bfloat16
T = 1297
# (label, (B, H, W, C), N, R, epilogue)
SHAPES = [
    ("neck 3x3 768->768 @36", (B, 36, 36, 768), 768, 3, "bias"), ("neck 3x3 768->768 @18", (B, 18, 18, 768), 768, 3, "bias"),
    ("fpn 3x3 256->256 @144", (B, 144, 144, 256), 256, 3, "bias"), ("fpn 3x3 256->256 @72", (B, 72, 72, 256), 256, 3, "bias"),
    ("fpn 3x3 256->256 @144 +resid", (B, 144, 144, 256), 256, 3, "resid"), ("psp bottleneck 3x3 1792->256 @18", (B, 18, 18, 1792), 256, 3, "bias"),
    ("eval: neck 3x3 768->768 @144 bn+relu", (B, 144, 144, 768), 768, 3, "bnrelu"),
    ("vit fc2 3072->768", (B, 1, T, 3072), 768, 1, "ls_resid"), ("dgrad taps 6912->768 @36", (B, 36, 36, 6912), 768, 1, "bias"),
    ("dgrad fuse taps 2304->256 @72", (B, 72, 72, 2304), 256, 1, "bias"), ("vit qkv", (1, 1, B * T, 768), 2304, 1, "bias"),
]


def timeit(fn, rounds=7, inner=4):
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


lib = _lib.load()
lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
print(f"batch {B}: us per call (TF/s); v3 = 256^2 ping-pong (8 waves), v4 = 3x3 shared staging (8 waves), v8 = 256^2 one wave per SIMD")
tot = {}
for label, shp, n, r, epi in SHAPES:
    x = torch.randn(shp, device="cuda").to(bf)
    w = (torch.randn(n, r * r * shp[3], device="cuda") * 0.05).to(bf)
    kw = dict(R=r, S=r, pad=r // 2, bias=torch.randn(n, device="cuda"))
    if epi == "resid":
        kw.update(resid=torch.randn(*shp[:3], n, device="cuda").to(bf))
    if epi == "bnrelu":
        kw.update(scale=torch.rand(n, device="cuda") + 0.5, shift=torch.randn(n, device="cuda"), act=ops.ACT_RELU)
    if epi == "ls_resid":
        kw.update(resid=torch.randn(*shp[:3], n, device="cuda"), scale=torch.full((n,), 1e-5, device="cuda"), out_dtype=torch.float32)
    M = shp[0] * shp[1] * shp[2]
    flops = 2 * M * n * r * r * shp[3]
    row = {}
    try:
        for v in (3, 4, 8, 3, 4, 8, -1):
            if v == 4 and r != 3:
                continue
            lib.gdl_debug_force_conv_variant(v)
            t = timeit(lambda: ops.conv_gemm(x, w, **kw))
            row[v] = min(row.get(v, 1e9), t)
    finally:
        lib.gdl_debug_force_conv_variant(-1)
    for v, t in row.items():
        tot[v] = tot.get(v, 0) + t
    print(f"  {label:38s} K {r * r * shp[3]:5d}: " + "  ".join(f"v{v} {t:6.1f} ({flops / t / 1e6:5.0f})" for v, t in row.items()), flush=True)
