#!/usr/bin/env python
"""Run ONE hot-path conv shape a few times (for rocprofv3 --pmc passes)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import ops  # noqa: E402
import ctypes, os  # noqa: E401,E402
from gdlhip import _lib  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
_l = _lib.load()
_l.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
_l.gdl_debug_force_conv_variant(int(os.environ.get("GDL_VARIANT", "-1")))
B = 8
x = torch.randn(B, 144, 144, 768, device="cuda").to(torch.bfloat16)
if which == "fwd":
    w = (torch.randn(768, 9 * 768, device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(3):
        ops.conv_gemm(x, w, R=3, S=3, pad=1)
else:
    dy = torch.randn(B, 144, 144, 768, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        ops.conv_wgrad(x, dy, R=3, S=3, pad=1)
torch.cuda.synchronize()
