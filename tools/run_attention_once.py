#!/usr/bin/env python
"""One launch set of the fused attention forward at the DOFA-base shape (batch 32, N = 1297, 12 heads) for PMC passes:
   tools/run_attention_once.py <kernel version: 3 | 4 | 13> [N] [B] [H]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

ver = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1297
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
H = int(sys.argv[4]) if len(sys.argv) > 4 else 12
lib = _lib.load()
lib.gdl_debug_set_flash_fwd.argtypes = [ctypes.c_int, ctypes.c_float]
lib.gdl_debug_set_flash_fwd(ver, 6.0)
qkv = torch.randn(B, N, 3 * H * 64, device="cuda").to(torch.bfloat16)
q, k, v = ops.split_qkv(qkv)
for _ in range(3):
    ops.attention_flash(q, k, v, H, return_lse=True)
torch.cuda.synchronize()
