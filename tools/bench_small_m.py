#!/usr/bin/env python
"""Tuning probe (round 6): the GEMM layers of DOFA-base + UperNet at the reference's per-GPU batch 4 (configs/dofa_config_RGB.yaml:85):
M = 5188 tokens / 5184 pixels of the 36^2 maps.  64^2 tile (variant 0, what the planner picks below 256 tiles of 128^2) against the
128^2 tile (variant 1), and both with four LDS stages (variants 12 / 11: the DMA of tile t+4 issued at K-step t) -- their results must be
bit-identical to the two-stage tiles (same MFMA order).  usage: bench_small_m.py [B]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

lib = _lib.load()
lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bf = torch.bfloat16
M = B * 1297
CASES = [("proj 768->768 f32 + resid", (1, 1, M), 768, 768, 1, "res"), ("fc2 3072->768 f32 + resid", (1, 1, M), 3072, 768, 1, "res"),
         ("qkv 768->2304", (1, 1, M), 768, 2304, 1, None), ("fc1 768->3072 gelu", (1, 1, M), 768, 3072, 1, "gelu"),
         ("neck 3x3 768->768 @36", (B, 36, 36), 768, 768, 3, None), ("lateral 1x1 768->256 @36", (B, 36, 36), 768, 256, 1, None),
         ("fpn 3x3 256->256 @36", (B, 36, 36), 256, 256, 3, None), ("neck taps 768->6912 @36", (B, 36, 36), 768, 6912, 1, None),
         ("neck 3x3 768->768 @18", (B, 18, 18), 768, 768, 3, None), ("fpn 3x3 256->256 @72", (B, 72, 72), 256, 256, 3, None),
         ("fpn taps 256->2304 @72", (B, 72, 72), 256, 2304, 1, None),
         ("ppm bottleneck 3x3 1792->256 @18", (B, 18, 18), 1792, 256, 3, None), ("its dgrad 3x3 256->1792 @18", (B, 18, 18), 256, 1792, 3, None),
         ("tap dgrad 6912->768 @36", (B, 36, 36), 6912, 768, 1, None), ("fpn tap dgrad 2304->256 @72", (B, 72, 72), 2304, 256, 1, None),
         ("lateral 1x1 768->256 @72", (B, 72, 72), 768, 256, 1, None), ("lateral dgrad 256->768 @72", (B, 72, 72), 256, 768, 1, None),
         ("neck 1x1 768->768 @36", (B, 36, 36), 768, 768, 1, None), ("fpn 3x3 256->256 @18", (B, 18, 18), 256, 256, 3, None)]


def timed(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, shp, k, n, rs, kind in CASES:
    x = torch.randn(*shp, k, device="cuda").to(bf)
    w = (torch.randn(n, rs * rs * k, device="cuda") * 0.02).to(bf)
    m = shp[0] * shp[1] * shp[2]
    kw = dict(R=rs, S=rs, pad=rs // 2)
    odt = bf
    if kind == "gelu":
        kw["act"] = ops.ACT_GELU
    if kind == "res":
        odt = torch.float32
        kw["resid"] = torch.randn(*shp, n, device="cuda")
        kw["scale"] = torch.full((n,), 1e-5, device="cuda")
        kw["shift"] = torch.zeros(n, device="cuda")
    out = torch.empty(*shp, n, device="cuda", dtype=odt)
    bias = torch.randn(n, device="cuda")
    VARIANTS = (-1, 0, 1, 12, 11, 13, 14)
    runs, outs = {v: [] for v in VARIANTS}, []
    for rnd in range(4):          # the variants interleaved, four rounds: the median is free of clock ramps and order effects
        for v in VARIANTS:
            lib.gdl_debug_force_conv_variant(v)
            runs[v].append(timed(lambda: ops.conv_gemm(x, w, bias=bias, out=out, **kw)))
            if rnd == 0:
                outs.append(out.clone())
    lib.gdl_debug_force_conv_variant(-1)
    res = [sorted(runs[v])[1] for v in VARIANTS]
    same = all(torch.equal(outs[1], o) for o in outs[2:]) and (kind == "gelu" or torch.equal(outs[0], outs[1]))
    fl = 2 * m * n * rs * rs * k
    print(f"{name:32s} M={m:6d}: planner {res[0]:6.1f} us | 64^2 {res[1]:6.1f} / 4 stages {res[3]:6.1f} us | 128^2 {res[2]:6.1f} / 4 stages {res[4]:6.1f} us | 128^2 by 8 waves 32x64 {res[5]:6.1f} / 64x32 {res[6]:6.1f} us"
          f" | best {fl / min(res) / 1e6:6.1f} TF/s | bit-identical {same}", flush=True)
