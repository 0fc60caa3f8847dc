#!/usr/bin/env python
"""Which tile should the planner pick at SMALL batch?  Times every forward conv_gemm shape of the DOFA + UperNet step at the given
per-GPU batch (default 4, the reference's) under each forced tile variant (0 = 64^2, 1 = 128^2, 3 = 256^2 ping-pong, 4 = 3x3
shared staging) and under the planner's own choice, interleaved in one process (HIP events, median of 5 x 4 calls).
   tools/bench_conv_variants.py [batch]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bf = torch.bfloat16
T = 1297
# (label, (B, H, W, C), N, R, f32 output + residual)
SHAPES = [
    ("vit qkv", (1, 1, B * T, 768), 2304, 1, False), ("vit proj", (B, 1, T, 768), 768, 1, True),
    ("vit fc1", (1, 1, B * T, 768), 3072, 1, False), ("vit fc2", (B, 1, T, 3072), 768, 1, True),
    ("neck 1x1 36", (B, 36, 36, 768), 768, 1, False), ("neck taps 36", (B, 36, 36, 768), 6912, 1, False),
    ("neck 3x3 36", (B, 36, 36, 768), 768, 3, False), ("neck 3x3 18", (B, 18, 18, 768), 768, 3, False),
    ("lateral 144", (B, 144, 144, 768), 256, 1, False), ("lateral 72", (B, 72, 72, 768), 256, 1, False),
    ("lateral 36", (B, 36, 36, 768), 256, 1, False), ("psp bottleneck 18", (B, 18, 18, 1792), 256, 3, False),
    ("fpn 3x3 144", (B, 144, 144, 256), 256, 3, False), ("fpn 3x3 72", (B, 72, 72, 256), 256, 3, False),
    ("fpn 3x3 36", (B, 36, 36, 256), 256, 3, False), ("fuse taps 72", (B, 72, 72, 256), 2304, 1, False),
    ("fuse taps 36", (B, 36, 36, 256), 2304, 1, False), ("fuse taps 18", (B, 18, 18, 256), 2304, 1, False),
    ("dgrad lateral 144", (B, 144, 144, 256), 768, 1, False), ("dgrad lateral 72", (B, 72, 72, 256), 768, 1, False),
    ("dgrad taps 36", (B, 36, 36, 6912), 768, 1, False), ("dgrad fuse taps 72", (B, 72, 72, 2304), 256, 1, False),
    ("dgrad fuse taps 36", (B, 36, 36, 2304), 256, 1, False),
]


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


lib = _lib.load()
lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
print(f"batch {B}: us per call under forced tile variants (planner's choice last)")
tot = {}
for label, shp, n, r, f32res in SHAPES:
    x = torch.randn(shp, device="cuda").to(bf)
    w = (torch.randn(n, r * r * shp[3], device="cuda") * 0.05).to(bf)
    kw = dict(R=r, S=r, pad=r // 2)
    if f32res:
        kw.update(resid=torch.randn(*shp[:3], n, device="cuda"), out_dtype=torch.float32)
    M = shp[0] * shp[1] * shp[2]
    flops = 2 * M * n * r * r * shp[3]
    row = {}
    for v in (0, 1, 3, 4, -1):
        if v == 4 and r != 3:
            continue
        lib.gdl_debug_force_conv_variant(v)
        row[v] = timeit(lambda: ops.conv_gemm(x, w, **kw))
    lib.gdl_debug_force_conv_variant(-1)
    best = min((t, v) for v, t in row.items() if v >= 0)
    for v, t in row.items():
        tot[v] = tot.get(v, 0) + (t if v != 4 else 0)
    tot["best"] = tot.get("best", 0) + best[0]
    print(f"  {label:20s} M {M:6d} N {n:5d} K {r * r * shp[3]:5d}: " + "  ".join(f"v{v} {t:6.1f}" for v, t in row.items() if v >= 0) +
          f"  | auto {row[-1]:6.1f} ({flops / row[-1] / 1e6:5.0f} TF/s)  best v{best[1]} {best[0]:6.1f} ({flops / best[0] / 1e6:5.0f} TF/s)", flush=True)
print(f"sum over the listed shapes: auto {tot[-1]:.0f} us, best-of-variants {tot['best']:.0f} us")
