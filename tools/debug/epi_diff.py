import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops
lib = _lib.load()
lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
lib.gdl_debug_set_conv_epilogue.argtypes = [ctypes.c_int]
B, T, K, N = 3, 1297, 128, 512
for dtype in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    xd = torch.randn(B, 1, T, K, device="cuda").to(dtype)
    wd = (torch.randn(N, K, device="cuda") * 0.1).to(dtype)
    bias = torch.randn(N, device="cuda")
    for odt in (torch.float32, torch.bfloat16):
        lib.gdl_debug_force_conv_variant(3)
        outs = []
        for v2 in (1, 2, 0):
            lib.gdl_debug_set_conv_epilogue(v2)
            outs.append(ops.conv_gemm(xd, wd, bias=bias, out_dtype=odt).float().view(B * T, N))
        lib.gdl_debug_set_conv_epilogue(1)
        lib.gdl_debug_force_conv_variant(-1)
        ref = (xd.float().view(B * T, K) @ wd.float().t() + bias)
        for nm, o in zip(("fixed", "runtime", "round2"), outs):
            d = (o - ref).abs()
            bad = d > 0.05 * ref.abs().max()
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print(dtype, odt, nm, "max err", d.max().item(), "bad elems", int(bad.sum()), "rows", rows[:12].tolist(), "n rows", len(rows), "cols", cols[:12].tolist(), "n cols", len(cols))
