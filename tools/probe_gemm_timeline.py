#!/usr/bin/env python
"""Tuning probe: where the time of a short-K GEMM goes.  Per block: shader cycles of prologue + K loop, of the epilogue,
and the block's start / end on the 100 MHz wall clock -> rounds per CU, gaps, effective clock.  Shapes: the ViT-base
linears and the 1x1 laterals of DOFA-base at per-GPU batch 32 (K = 768: 12 K-steps) next to the deep-K neck conv."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

lib = _lib.load()
lib.gdl_debug_set_conv_probe.argtypes = [ctypes.c_void_p]
lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
lib.gdl_debug_set_conv_dbg.argtypes = [ctypes.c_int]
lib.gdl_debug_set_conv_epilogue.argtypes = [ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bf = torch.bfloat16
M = B * 1297
SHAPES = [("qkv 768->2304 bf16 out", M, 768, 2304, None, bf), ("proj 768->768 f32 out + resid", M, 768, 768, "res", torch.float32),
          ("fc1 768->3072 gelu", M, 768, 3072, "gelu", bf), ("fc2 3072->768 f32 + resid", M, 3072, 768, "res", torch.float32),
          ("lateral 768->256 @144", B * 144 * 144, 768, 256, None, bf)]
buf = torch.zeros(12288, device="cuda", dtype=torch.int64)


def run(name, m, k, n, kind, odt, variant, dbg):
    conv = isinstance(m, tuple)            # (B, H, W): 3x3 / pad 1 convolution on an NHWC map
    if conv:
        x = torch.randn(*m, k, device="cuda").to(bf)
        w = (torch.randn(n, 9 * k, device="cuda") * 0.02).to(bf)
        m = m[0] * m[1] * m[2]
    else:
        x = torch.randn(1, 1, m, k, device="cuda").to(bf)
        w = (torch.randn(n, k, device="cuda") * 0.05).to(bf)
    bias = torch.randn(n, device="cuda")
    out = torch.empty(*(x.shape[:3] if conv else (1, 1, m)), n, device="cuda", dtype=odt)
    kw = dict(R=3, S=3, pad=1) if conv else {}
    kk = 9 * k if conv else k
    if kind == "gelu":
        kw["act"] = ops.ACT_GELU
    if kind == "res":
        kw["resid"] = torch.randn(1, 1, m, n, device="cuda")
        kw["scale"] = torch.full((n,), 1e-5, device="cuda")
        kw["shift"] = torch.zeros(n, device="cuda")
    lib.gdl_debug_force_conv_variant(variant)
    lib.gdl_debug_set_conv_dbg(dbg)
    fn = lambda: ops.conv_gemm(x, w, bias=bias, out=out, **kw)  # noqa: E731
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    buf.zero_()
    lib.gdl_debug_set_conv_probe(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.gdl_debug_set_conv_probe(None)
    lib.gdl_debug_set_conv_dbg(0)
    lib.gdl_debug_force_conv_variant(-1)
    tm = 256 if variant in (2, 3, 4, 6, 8, 9, 10) else 128
    tn = 128 if variant == 6 else tm
    nb = min(2048, ((m + tm - 1) // tm) * ((n + tn - 1) // tn))
    kl = buf[:4096].view(2048, 2)[:nb].double().cpu()
    tot = buf[4096:4096 + nb].double().cpu()
    tl = buf[8192:12288].view(2048, 2)[:nb].double().cpu()
    t0 = tl[:, 0].min()
    dur = (tl[:, 1] - tl[:, 0]) / 100.0          # us per block
    span = (tl[:, 1].max() - t0) / 100.0
    mhz = (kl[:, 0] / (kl[:, 1] / 100.0)).median().item()
    first = (tl[:, 0] - t0).sort().values / 100.0
    kt = kk // 64
    if dbg == 7:
        ph = buf[10240:10240 + 64].view(8, 8)[:, :5].double().cpu() / kt
        names = ("groups 0-2", "wait own DMA", "barrier", "DMA issue", "group 3")
        for w in range(8):
            print(f"    wave {w}: " + ", ".join(f"{n} {ph[w, i]:.0f}" for i, n in enumerate(names)) + f"  (sum {ph[w].sum():.0f} cyc/K-step)")
    print(f"{name:32s} v{variant} dbg{dbg}: {2 * m * kk * n / us / 1e6:7.1f} TF/s ({us:6.0f} us); blocks {nb}; "
          f"prologue+K loop {kl[:, 0].median():.0f} cyc ({kl[:, 0].median() / kt:.0f}/K-step), epilogue "
          f"{(tot - kl[:, 0]).median():.0f} cyc; block {dur.median():.1f} us (p10 {dur.quantile(0.1):.1f} p90 {dur.quantile(0.9):.1f}); "
          f"probe span {span:.0f} us; clock {mhz:.0f} MHz; start of block #256/#512/#1024: "
          f"{first[min(256, nb - 1)]:.1f}/{first[min(512, nb - 1)]:.1f}/{first[min(1024, nb - 1)]:.1f} us", flush=True)


FULL = len(sys.argv) > 2 and sys.argv[2] == "full"
if len(sys.argv) > 2 and sys.argv[2] == "dual":   # 256^2 ping-pong (one workgroup per CU) vs the dual-resident 256 x 128 tile
    for shp in SHAPES:
        for variant, dbg in ((3, 0), (8, 0), (8, 1)):
            run(*shp, variant, dbg)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "persist":   # one-tile 256^2 ping-pong workgroups vs the persistent form (dbg 11: without the prefetch under the epilogue)
    K256 = [("dgrad lateral 256->768 @144", B * 144 * 144, 256, 768, None, bf), ("fuse taps 256->2304 @72", B * 72 * 72, 256, 2304, None, bf),
            ("neck taps 768->6912 @36", B * 36 * 36, 768, 6912, None, bf)]
    for shp in SHAPES[:2] + [SHAPES[4]] + K256:
        for variant, dbg in ((3, 0), (9, 0), (9, 11)):
            run(*shp, variant, dbg)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "parked":   # parked tile (10: as is / 20: phase-2 stores dropped / 21: no phase 1) vs the persistent 8-wave tile and w4
    K256 = [("dgrad lateral 256->768 @144", B * 144 * 144, 256, 768, None, bf), ("fuse taps 256->2304 @72", B * 72 * 72, 256, 2304, None, bf)]
    for shp in [SHAPES[0], SHAPES[4], ("neck taps 768->6912 @36", B * 36 * 36, 768, 6912, None, bf)]:
        for variant, dbg in ((9, 0), (8, 0), (10, 0), (10, 20), (10, 21)):
            run(*shp, variant, dbg)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "epilogue_parts":   # round-3 epilogue: as is / without its global stores (10) / without residual loads (11)
    for shp in (SHAPES[0], SHAPES[1]):
        for dbg in (0, 10, 11, 0, 10, 11):
            run(*shp, 3, dbg)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "epilogue":      # round-3 epilogue (1) against the round-2 one (0), 256^2 ping-pong tiles
    for shp in SHAPES:
        for v2 in (0, 1, 0, 1):
            lib.gdl_debug_set_conv_epilogue(v2)
            print(f"epilogue v{2 if v2 else 1}: ", end="")
            run(*shp, 3, 0)
    lib.gdl_debug_set_conv_epilogue(1)
    sys.exit(0)
for shp in ([] if len(sys.argv) > 2 and sys.argv[2] in ("convs", "phases") else SHAPES):
    for variant in (3, 1):
        for dbg in ((0, 1, 2) if FULL else (0,)):
            if variant == 1 and dbg:
                continue
            run(*shp, variant, dbg)
CONVS = [("neck 3x3 768->768 @144", (B, 144, 144), 768, 768, None, bf), ("fusion 3x3 1024->256 @144", (B, 144, 144), 1024, 256, None, bf),
         ("fpn 3x3 256->256 @144", (B, 144, 144), 256, 256, None, bf), ("neck 3x3 768->768 @72", (B, 72, 72), 768, 768, None, bf)]
if len(sys.argv) > 2 and sys.argv[2] == "phases":
    run(*SHAPES[0], 3, 7)
    run(*SHAPES[3], 3, 7)
    run(*CONVS[0], 3, 7)
    sys.exit(0)
for shp in CONVS:
    run(*shp, 4, 0)
    run(*shp, 3, 0)
    if FULL:
        run(*shp, 4, 1)
        run(*shp, 4, 2)
