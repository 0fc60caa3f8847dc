#!/usr/bin/env python
"""GPU micro-benchmark of the fused attention kernels at the DOFA shapes: forward v1 (separate V^T pass) vs v2, and the
fused backward vs the materialised one.  TF/s = algorithmic flops (4 N^2 D per head forward, 2.5x that backward)."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402

lib = _lib.load()
lib.gdl_debug_set_flash_fwd.argtypes = [ctypes.c_int, ctypes.c_float]

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
    # forward schedules: 2 = round 2 (S, softmax, PV for both query tiles of a wave at once), 3 = staggered query tiles,
    # 3 + deferred running maximum (threshold in the exp2 domain)
    res = {}
    # 2 = round 2 (64-query waves), 3 = round 3 (32-query waves; exact / deferred running maximum)
    # 4 / 5 = round 5: S of tile t+1 issued before the softmax of tile t (Q fragments in registers / re-read from the LDS)
    for tag, ver, defer in (("v2", 2, 0.0), ("v3", 3, 0.0), ("v3 defer 6", 3, 6.0), ("v4 defer 6", 4, 6.0), ("v5 defer 6", 5, 6.0),
                            ("v3 defer 6 again", 3, 6.0), ("v4 defer 6 again", 4, 6.0), ("v5 defer 6 again", 5, 6.0)):
        lib.gdl_debug_set_flash_fwd(ver, defer)
        res[tag] = (timeit(lambda: ops.attention_flash(q, k, v, H, return_lse=True)), ops.attention_flash(q, k, v, H, return_lse=True))
    lib.gdl_debug_set_flash_fwd(-1, 6.0)
    ref = torch.nn.functional.scaled_dot_product_attention(*(t.view(B, N, H, 64).transpose(1, 2).float() for t in (q, k, v)))
    ref = ref.transpose(1, 2).reshape(B, N, D)
    print("   forward schedules: " + " | ".join(
        f"{tag} {t:6.0f} us = {fl / t / 1e6:5.0f} TF/s, max err vs f32 {(o_.float() - ref).abs().max().item():.2e}"
        + ("" if tag == "v2" else f", == v2: {torch.equal(o_, res['v2'][1][0])}") for tag, (t, (o_, _)) in res.items()), flush=True)
    t2 = res["v3 defer 6"][0]
    o, lse = ops.attention_flash(q, k, v, H, return_lse=True)
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = ops.split_qkv(dqkv)
    t3 = timeit(lambda: ops.attention_bwd(q, k, v, do, H, dq, dk, dv, o=o, lse=lse), rounds=3, inner=2)
    line = (f"{name}: fwd v1 (+V^T pass) {t1:7.0f} us = {fl / t1 / 1e6:6.1f} TF/s | fwd v3 {t2:7.0f} us = {fl / t2 / 1e6:6.1f} TF/s | "
            f"fused bwd {t3:7.0f} us = {2.5 * fl / t3 / 1e6:6.1f} TF/s")
    if N <= 2048 and B * H * N * N * 2 * 2 < 8e9:
        t4 = timeit(lambda: ops.attention_bwd(q, k, v, do, H, dq, dk, dv), rounds=3, inner=2)
        line += f" | materialised bwd {t4:7.0f} us"
    print(line, flush=True)
