#!/usr/bin/env python
"""Short-K GEMM class of the DOFA + UperNet step (bench.py's roofline.by_layer shapes, batch 32, bf16): the 256^2 ping-pong tile
(variant 3, one workgroup per CU) against the dual-resident 256 x 128 tile (variant 6, two workgroups per CU), interleaved in one
process with each layer's real epilogue.  HIP events, median of 7 x 6 calls.
   tools/bench_dual.py [batch] [dbg modes for variant 6, e.g. 0 8]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DBG = [int(v) for v in sys.argv[2:]] or [0]
bf = torch.bfloat16
T = 1297
# (label, M, K, N, epilogue): "bf16" = bias -> bf16, "gelu" = bias + GELU -> bf16, "res" = LayerScale * (acc + bias) + f32 residual -> f32
SHAPES = [
    ("vit fc1", B * T, 768, 3072, "gelu"), ("vit fc2", B * T, 3072, 768, "res"), ("vit qkv", B * T, 768, 2304, "bf16"),
    ("vit proj", B * T, 768, 768, "res"), ("neck taps 36", B * 1296, 768, 6912, "bf16"), ("dgrad taps 36", B * 1296, 6912, 768, "bf16"),
    ("dgrad lateral 144", B * 144 * 144, 256, 768, "bf16"), ("lateral 144", B * 144 * 144, 768, 256, "bf16"),
    ("fuse taps 72", B * 72 * 72, 256, 2304, "bf16"), ("dgrad fuse taps 72", B * 72 * 72, 2304, 256, "bf16"),
    ("neck 1x1 36", B * 1296, 768, 768, "bf16"), ("dgrad lateral 72", B * 72 * 72, 256, 768, "bf16"),
    ("lateral 72", B * 72 * 72, 768, 256, "bf16"),
]


def timeit(fn, rounds=7, inner=6):
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
lib.gdl_debug_set_conv_dbg.argtypes = [ctypes.c_int]
lib.gdl_debug_set_conv_epilogue.argtypes = [ctypes.c_int]
lib.gdl_debug_set_conv_ngroup_kb.argtypes = [ctypes.c_int]
print(f"batch {B}: us per call (TF/s); v3 = 256^2 ping-pong (8 waves), v6 = dual-resident 256x128, v8 = 256^2 one wave per SIMD, v9 = persistent 256^2 ping-pong (/d0.5 = second measurement)" + "".join(f", v6/dbg{d}" for d in DBG if d))
tot = {}
for label, M, K, N, epi in SHAPES:
    x = torch.randn(1, 1, M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    bias = torch.randn(N, device="cuda")
    kw = dict(bias=bias)
    if epi == "gelu":
        kw.update(act=ops.ACT_GELU)
    if epi == "res":
        kw.update(resid=torch.randn(1, 1, M, N, device="cuda"), scale=torch.full((N,), 1e-5, device="cuda"), out_dtype=torch.float32)
    out = torch.empty(1, 1, M, N, device="cuda", dtype=torch.float32 if epi == "res" else bf)
    flops = 2 * M * N * K
    row = {}
    try:
        # dbg "100": element-wise terms decided at run time (the round-3 epilogue); "200": N tiles not grouped (round-3 tile order)
        for v, d in [(3, 0)] + [(6, d) for d in DBG] + [(8, 0), (9, 0), (3, 0.5), (9, 0.5), (-1, 0)]:
            lib.gdl_debug_force_conv_variant(v)
            lib.gdl_debug_set_conv_dbg(int(d) if d < 100 else 0)
            lib.gdl_debug_set_conv_epilogue(2 if d == 100 else 1)
            lib.gdl_debug_set_conv_ngroup_kb(0 if d in (100, 200) else 2560)
            row[v, d] = timeit(lambda: ops.conv_gemm(x, w, out=out, **kw))
    finally:
        lib.gdl_debug_force_conv_variant(-1)
        lib.gdl_debug_set_conv_dbg(0)
        lib.gdl_debug_set_conv_epilogue(1)
        lib.gdl_debug_set_conv_ngroup_kb(2560)
    for key, t in row.items():
        tot[key] = tot.get(key, 0) + t
    print(f"  {label:20s} M {M:6d} N {N:5d} K {K:5d} {epi:5s}: " +
          "  ".join(f"v{v}{'/d%s' % d if d else ''} {t:6.1f} ({flops / t / 1e6:5.0f})" for (v, d), t in row.items()), flush=True)
print("sum: " + "  ".join(f"v{v}{'/d%s' % d if d else ''} {t:.0f} us" for (v, d), t in tot.items()))
