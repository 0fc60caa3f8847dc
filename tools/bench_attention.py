#!/usr/bin/env python
"""GPU micro-benchmark of the fused attention kernels at the DOFA shapes: forward v1 (separate V^T pass) vs v2, and the
fused backward vs the materialised one.  TF/s = algorithmic flops (4 N^2 D per head forward, 2.5x that backward)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import ops  # noqa: E402

bf = torch.bfloat16


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
    return ts[len(ts) // 2] * 1e3   # us


for name, B, N, H in (("DOFA-base 512^2 (N=1297, 12 heads, batch 32)", 32, 1297, 12),
                      ("DOFA-large 1024^2 (N=5330, 16 heads, batch 2)", 2, 5330, 16),
                      ("DOFA-base batch 4", 4, 1297, 12)):
    D = H * 64
    qkv = torch.randn(B, N, 3 * D, device="cuda").to(bf)
    q, k, v = ops.split_qkv(qkv)
    fl = 4.0 * N * N * 64 * H * B
    t1 = timeit(lambda: ops.attention_flash_v1(q, k, v, H))
    t2 = timeit(lambda: ops.attention_flash(q, k, v, H, return_lse=True))
    o, lse = ops.attention_flash(q, k, v, H, return_lse=True)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = ops.split_qkv(dqkv)
    t3 = timeit(lambda: ops.attention_bwd(q, k, v, do, H, dq, dk, dv, o=o, lse=lse), rounds=3, inner=2)
    line = (f"{name}: fwd v1 (+V^T pass) {t1:7.0f} us = {fl / t1 / 1e6:6.1f} TF/s | fwd v2 {t2:7.0f} us = {fl / t2 / 1e6:6.1f} TF/s | "
            f"fused bwd {t3:7.0f} us = {2.5 * fl / t3 / 1e6:6.1f} TF/s")
    if N <= 2048 and B * H * N * N * 2 * 2 < 8e9:
        t4 = timeit(lambda: ops.attention_bwd(q, k, v, do, H, dq, dk, dv), rounds=3, inner=2)
        line += f" | materialised bwd {t4:7.0f} us"
    print(line, flush=True)
