#!/usr/bin/env python
"""The parked-tile kernel (variant 10, conv_gemm_w4p.hip) against the tiles it replaces, per layer of the DOFA + UperNet step at
batch 32: 9 = persistent 256^2 ping-pong, 8 = 256^2 one wave per SIMD, 6 = dual-resident 256 x 128, -1 = the planner's choice.
Interleaved in one process, HIP events, median of 7 x 4 calls, random operands.
   tools/bench_w4p.py [batch]"""
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
# (label, (B, H, W, C), N, kind): kind "bf16" bias only; "res" f32 + LayerScale + DropPath + residual (pre_bf16); "bnrelu"
SHAPES = [
    ("vit qkv", (B, 1, T, 768), 2304, "bf16"), ("vit proj", (B, 1, T, 768), 768, "res"), ("vit fc2", (B, 1, T, 3072), 768, "res"),
    ("neck 1x1 36", (B, 36, 36, 768), 768, "bf16"), ("neck taps 36", (B, 36, 36, 768), 6912, "bf16"),
    ("dgrad taps 36", (B, 36, 36, 6912), 768, "bf16"),
    ("lateral 144", (B, 144, 144, 768), 256, "bf16"), ("lateral 144 eval", (B, 144, 144, 768), 256, "bnrelu"),
    ("lateral 72", (B, 72, 72, 768), 256, "bf16"),
    ("dgrad lateral 144", (B, 144, 144, 256), 768, "bf16"), ("dgrad lateral 72", (B, 72, 72, 256), 768, "bf16"),
    ("fuse taps 72", (B, 72, 72, 256), 2304, "bf16"), ("dgrad fuse taps 72", (B, 72, 72, 2304), 256, "bf16"),
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
print(f"batch {B}: us per call (TF/s); v9 persistent 8-wave, v8 one wave per SIMD, v6 dual-resident, v10 parked tile, v-1 planner")
tot = {}
for label, shp, n, kind in SHAPES:
    x = torch.randn(shp, device="cuda").to(bf)
    w = (torch.randn(n, shp[3], device="cuda") * 0.05).to(bf)
    bias = torch.randn(n, device="cuda")
    kw = dict(bias=bias)
    if kind == "res":
        kw.update(resid=torch.randn(*shp[:3], n, device="cuda"), out=torch.empty(*shp[:3], n, device="cuda"),
                  scale=torch.randn(n, device="cuda"), batch_scale=torch.ones(shp[0], device="cuda"))
    elif kind == "bnrelu":
        kw.update(scale=torch.randn(n, device="cuda"), shift=torch.randn(n, device="cuda"), act=ops.ACT_RELU)
    M = shp[0] * shp[1] * shp[2]
    flops = 2 * M * n * shp[3]
    row = {}
    for rep in range(2):
        for v in (9, 8, 6, 10, -1):
            lib.gdl_debug_force_conv_variant(v)
            t = timeit(lambda: ops.conv_gemm(x, w, **kw))
            row[v] = min(row.get(v, 1e9), t)
    lib.gdl_debug_force_conv_variant(-1)
    for v, t in row.items():
        tot[v] = tot.get(v, 0) + t
    print(f"  {label:20s} M {M:6d} N {n:5d} K {shp[3]:5d} {kind:6s}: " +
          "  ".join(f"v{v} {t:6.1f} ({flops / t / 1e6:5.0f})" for v, t in row.items()), flush=True)
print("sum: " + "  ".join(f"v{v} {t:.0f} us" for v, t in tot.items()))
