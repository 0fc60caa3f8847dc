#!/usr/bin/env python
"""GPU micro-benchmark: 128^2 (variant 1) vs 256^2 ping-pong (variant 3) tiles on the ViT-base linears and the 1x1
decoder convs of DOFA-base at per-GPU batch 32 (short K: 12 K-steps), bf16, interleaved in one process."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DEV, bf = "cuda", torch.bfloat16
M = B * 1297
LAYERS = [("vit qkv 768->2304", M, 768, 2304), ("vit proj 768->768", M, 768, 768), ("vit fc1 768->3072", M, 768, 3072),
          ("vit fc2 3072->768", M, 3072, 768), ("lateral 1x1 768->256 @144", B * 144 * 144, 768, 256),
          ("neck lateral 768->768 @36", B * 36 * 36, 768, 768)]


def timeit(fn, rounds=5, inner=4):
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
    return ts[len(ts) // 2]


def main():
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    print(f"batch {B}: TF/s (us)  auto | 128^2 | 256^2 ping-pong")
    for name, m, k, n in LAYERS:
        x = torch.randn(1, 1, m, k, device=DEV).to(bf)
        w = (torch.randn(n, k, device=DEV) * 0.05).to(bf)
        bias = torch.randn(n, device=DEV)
        out = torch.empty(1, 1, m, n, device=DEV, dtype=bf)
        flops = 2 * m * k * n
        res = []
        for v in (-1, 1, 3):
            lib.gdl_debug_force_conv_variant(v)
            try:
                t = timeit(lambda: ops.conv_gemm(x, w, bias=bias, act=ops.ACT_GELU, out=out))
            finally:
                lib.gdl_debug_force_conv_variant(-1)
            res.append(f"{flops / t / 1e9:7.1f} ({t * 1e3:5.0f})")
        print(f"{name:28s} GF {flops / 1e9:7.1f}  " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
