#!/usr/bin/env python
"""GPU micro-benchmark of the MFMA kernels on the hot path's real shapes (per-GPU batch 8):
conv_gemm tile variants A/B (interleaved rounds in ONE process), wgrad, flash attention."""

import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = "cuda"
bf = torch.bfloat16


def timeit(fn, rounds=5, inner=4):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / inner)
    best.sort()
    return best[len(best) // 2], best[0]


CONVS = [  # name, (B,H,W,C), N, R
    ("neck3x3_144", (B, 144, 144, 768), 768, 3),
    ("neck3x3_72", (B, 72, 72, 768), 768, 3),
    ("fuse3x3_144", (B, 144, 144, 1024), 256, 3),
    ("fpn3x3_144", (B, 144, 144, 256), 256, 3),
    ("lat1x1_144", (B, 144, 144, 768), 256, 1),
    ("dgrad_fuse", (B, 144, 144, 256), 1024, 3),
    ("vit_qkv", (1, 1, B * 1297, 768), 2304, 1),
    ("vit_fc1", (1, 1, B * 1297, 768), 3072, 1),
    ("vit_fc2", (1, 1, B * 1297, 3072), 768, 1),
    ("vit_proj", (1, 1, B * 1297, 768), 768, 1),
]


def main():
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    lib.gdl_debug_set_conv_korder.argtypes = [ctypes.c_int]
    lib.gdl_debug_force_wgrad_small.argtypes = [ctypes.c_int]
    lib.gdl_debug_set_conv_dbg.argtypes = [ctypes.c_int]
    print(f"batch {B}")
    for name, shp, n, r in CONVS:
        x = torch.randn(shp, device=DEV).to(bf)
        w = (torch.randn(n, r * r * shp[3], device=DEV) * 0.05).to(bf)
        flops = 2 * shp[0] * shp[1] * shp[2] * n * r * r * shp[3]
        row, outs = [], {}
        lib.gdl_debug_set_conv_korder(0)
        lib.gdl_debug_force_conv_variant(2 if n % 256 == 0 else 1)
        med0, _ = timeit(lambda: ops.conv_gemm(x, w, R=r, S=r, pad=r // 2))
        lib.gdl_debug_set_conv_korder(1)
        lib.gdl_debug_set_conv_dbg(1)
        med1, _ = timeit(lambda: ops.conv_gemm(x, w, R=r, S=r, pad=r // 2))
        lib.gdl_debug_set_conv_dbg(2)
        med2, _ = timeit(lambda: ops.conv_gemm(x, w, R=r, S=r, pad=r // 2))
        lib.gdl_debug_set_conv_dbg(0)
        print(f"   {name}: compute-only (no DMA) {flops / med1 / 1e9:7.1f} TF/s-equivalent, load-only (no MFMA) {flops / med2 / 1e9:7.1f}")
        for v in (1, 2, 3, 4, -1):
            if v == 4 and (r != 3 or n % 256):
                row.append("   -   ")
                continue
            if v in (2, 3) and n % 256:
                row.append("   -   ")
                continue
            lib.gdl_debug_force_conv_variant(v)
            med, mn = timeit(lambda: ops.conv_gemm(x, w, R=r, S=r, pad=r // 2))
            outs[v] = ops.conv_gemm(x, w, R=r, S=r, pad=r // 2)
            row.append(f"{flops / med / 1e9:7.1f}")
        lib.gdl_debug_force_conv_variant(-1)
        same = "" if 3 not in outs else f"  v3==v2: {torch.equal(outs[2], outs[3])}"
        if 4 in outs:
            same += f" v4 max|diff| {(outs[4].float() - outs[3].float()).abs().max().item():.2e}"
        print(f"conv_gemm {name:14s} GF {flops / 1e9:8.1f}  TF/s v1(128^2) {row[0]}  v2(256^2) {row[1]}  "
              f"v3(256^2 ping-pong) {row[2]}  v4(3x3 shared staging) {row[3]}  auto {row[4]}{same}  [tap-outer K order: {flops / med0 / 1e9:7.1f}]")
    for name, shp, n, r in CONVS[:6]:
        x = torch.randn(shp, device=DEV).to(bf)
        dy = torch.randn((shp[0], shp[1], shp[2], n), device=DEV).to(bf)
        flops = 2 * shp[0] * shp[1] * shp[2] * n * r * r * shp[3]
        lib.gdl_debug_force_wgrad_small(1)
        med_s, _ = timeit(lambda: ops.conv_wgrad(x, dy, R=r, S=r, pad=r // 2), rounds=3, inner=2)
        ref = ops.conv_wgrad(x, dy, R=r, S=r, pad=r // 2)
        lib.gdl_debug_force_wgrad_small(0)
        med, mn = timeit(lambda: ops.conv_wgrad(x, dy, R=r, S=r, pad=r // 2), rounds=3, inner=2)
        got = ops.conv_wgrad(x, dy, R=r, S=r, pad=r // 2)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        print(f"wgrad     {name:14s} GF {flops / 1e9:8.1f}  TF/s {flops / med / 1e9:7.1f}  ({med * 1e3:.0f} us)  "
              f"[128^2 kernel only: {flops / med_s / 1e9:7.1f}; max rel diff {err:.1e}]")
    qkv = torch.randn(B, 1297, 2304, device=DEV).to(bf)
    flops = 4 * 1297 * 1297 * 768 * B
    med, mn = timeit(lambda: ops.attention_flash(*ops.split_qkv(qkv), 12))
    print(f"flash_attn B{B} N1297 H12     GF {flops / 1e9:8.1f}  TF/s {flops / med / 1e9:7.1f}  ({med * 1e3:.0f} us)")


if __name__ == "__main__":
    main()
