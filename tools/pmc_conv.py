#!/usr/bin/env python
"""Run ONE hot-path shape a few times (for rocprofv3 --pmc passes).
usage: pmc_conv.py {fwd|wgrad|qkv|fc2} [batch]   (GDL_VARIANT=n forces a conv_gemm tile variant)"""
import ctypes
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
_l = _lib.load()
_l.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
_l.gdl_debug_force_conv_variant(int(os.environ.get("GDL_VARIANT", "-1")))
bf = torch.bfloat16
if which in ("fwd", "wgrad"):
    x = torch.randn(B, 144, 144, 768, device="cuda").to(bf)
    if which == "fwd":
        w = (torch.randn(768, 9 * 768, device="cuda") * 0.05).to(bf)
        for _ in range(3):
            ops.conv_gemm(x, w, R=3, S=3, pad=1)
    else:
        dy = torch.randn(B, 144, 144, 768, device="cuda").to(bf)
        for _ in range(3):
            ops.conv_wgrad(x, dy, R=3, S=3, pad=1)
else:
    k, n = (768, 2304) if which == "qkv" else (3072, 768)
    x = torch.randn(1, 1, B * 1297, k, device="cuda").to(bf)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(bf)
    bias = torch.randn(n, device="cuda")
    for _ in range(3):
        ops.conv_gemm(x, w, bias=bias)
torch.cuda.synchronize()
