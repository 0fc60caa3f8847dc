#!/usr/bin/env python
"""Tuning probe: shader cycles and wall time of the conv_gemm K loop per block (effective clock, cycles per K-step)."""
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
B = 8
x = torch.randn(B, 144, 144, 768, device="cuda").to(torch.bfloat16)
w = (torch.randn(768, 9 * 768, device="cuda") * 0.05).to(torch.bfloat16)
buf = torch.zeros(8192, device="cuda", dtype=torch.int64)
KT = 9 * 768 // 64
for variant in (3, 4):
    for dbg in (0, 1, 2):
        lib.gdl_debug_force_conv_variant(variant)
        lib.gdl_debug_set_conv_dbg(dbg)
        for _ in range(3):
            ops.conv_gemm(x, w, R=3, S=3, pad=1)
        lib.gdl_debug_set_conv_probe(buf.data_ptr())
        ops.conv_gemm(x, w, R=3, S=3, pad=1)
        torch.cuda.synchronize()
        lib.gdl_debug_set_conv_probe(None)
        tot = buf[4096:4096 + 1944].double().cpu()
        v = buf[:4096].view(2048, 2)[:1944].double().cpu()
        cyc, ticks = v[:, 0], v[:, 1]
        mhz = (cyc / (ticks / 100.0)).median().item()
        print(f"variant {variant} dbg {dbg}: K-loop cycles/block median {cyc.median().item():.0f} "
              f"(min {cyc.min().item():.0f} max {cyc.max().item():.0f}) = {cyc.median().item() / KT:.0f} per K-step; "
              f"shader clock ~{mhz:.0f} MHz; K loop + epilogue {tot.median().item():.0f} cycles "
              f"(epilogue {tot.median().item() - cyc.median().item():.0f})")
lib.gdl_debug_set_conv_dbg(0)
lib.gdl_debug_force_conv_variant(-1)
