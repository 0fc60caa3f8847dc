#!/usr/bin/env python
"""One launch of each gather-sum variant on the neck's x4 level at batch 32 (for rocprofv3 --pmc): mode from argv[1]."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

lib = _lib.load()
lib.gdl_debug_set_tapsum_mfma.argtypes = [ctypes.c_int]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
z = (torch.randn(32, 36, 36, 9 * 768, device="cuda") * 0.5).to(torch.bfloat16)
lib.gdl_debug_set_tapsum_mfma(mode)
for _ in range(3):
    y = ops.resize_conv3x3_fwd_sum([z], (144, 144))
torch.cuda.synchronize()
