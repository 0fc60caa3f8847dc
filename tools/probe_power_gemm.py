#!/usr/bin/env python
"""Helper of tools/probe_power_clock.sh: loops ONE kernel for ~3 s and prints its rate -- the neck's tap-product GEMM
(41472 x 6912 x 768, bf16) on random or on zero operands, or a bf16 BatchNorm-apply pass over a 1 GB map (HBM streaming)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import ops  # noqa: E402

mode = sys.argv[1]
bf = torch.bfloat16
if mode in ("random", "zero"):
    x = (torch.randn(1, 1, 41472, 768, device="cuda") if mode == "random" else torch.zeros(1, 1, 41472, 768, device="cuda")).to(bf)
    w = ((torch.randn(6912, 768, device="cuda") * 0.05) if mode == "random" else torch.zeros(6912, 768, device="cuda")).to(bf)
    out = torch.empty(1, 1, 41472, 6912, device="cuda", dtype=bf)
    fn, work, unit = (lambda: ops.conv_gemm(x, w, out=out)), 2 * 41472 * 6912 * 768 / 1e12, "TF/s"
else:
    x = torch.randn(663552, 768, device="cuda").to(bf)
    y = torch.empty_like(x)
    fn, work, unit = (lambda: y.copy_(x)), 2 * x.numel() * 2 / 1e12, "TB/s"
for _ in range(5):
    fn()
torch.cuda.synchronize()
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n += 20
dt = time.perf_counter() - t0
print(f"{mode}: {n} calls, {1e6 * dt / n:.1f} us per call, {work * n / dt:.2f} {unit}")
