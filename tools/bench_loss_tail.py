#!/usr/bin/env python
"""The loss tail of one head of the DOFA training step (batch 32, 5 classes, 144^2 -> 512^2): materialised path (upsample the
logits, Dice forward, Dice backward, transposed upsample) vs the low-resolution path (round 5), us per call."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import ops  # noqa: E402

B, K, h, H = 32, 5, 144, 512
low = torch.randn(B, h, h, K, device="cuda") * 2
tgt = torch.randint(0, K, (B, H, H), device="cuda")
up = torch.tensor(0.4, device="cuda")


def timeit(fn, rounds=5, inner=5):
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
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return sorted(ts)[len(ts) // 2]


full = ops.upsample_logits(low, (H, H))
_, sums = ops.dice_loss_fwd(full, tgt)
g = ops.dice_loss_bwd(full, tgt, sums, up)
print(f"materialised: upsample {timeit(lambda: ops.upsample_logits(low, (H, H))):6.1f} | dice fwd {timeit(lambda: ops.dice_loss_fwd(full, tgt)):6.1f} | "
      f"dice bwd {timeit(lambda: ops.dice_loss_bwd(full, tgt, sums, up)):6.1f} | upsample bwd {timeit(lambda: ops.upsample_logits_bwd(g, (h, h))):6.1f} us")
print(f"low-res     : fwd {timeit(lambda: ops.dice_loss_lowres_fwd(low, tgt, (H, H))):6.1f} | bwd {timeit(lambda: ops.dice_loss_lowres_bwd(low, tgt, (H, H), sums, up)):6.1f} us")

# the 1x1 classifier on the decoder output [32, 144, 144, 256] bf16 (340 MB), with and without a Dropout2d channel scale
feat = torch.randn(B, h, h, 256, device="cuda").to(torch.bfloat16)
w, bias = torch.randn(K, 256, device="cuda") * 0.05, torch.randn(K, device="cuda")
cs = (torch.rand(B, 256, device="cuda") < 0.9).float() / 0.9
dlog = torch.randn(B, h, h, K, device="cuda")
nb = feat.numel() * 2
for name, scale in (("main head", None), ("aux head (Dropout2d scale)", cs)):
    t_f = timeit(lambda: ops.head_1x1(feat, w, bias, scale))
    t_b = timeit(lambda: ops.head_1x1_bwd(feat, dlog, w, scale))
    print(f"head 1x1, {name:27s}: fwd {t_f:6.1f} us ({nb / t_f / 1e3:6.0f} GB/s) | bwd (dfeat + dw partials + final) {t_b:6.1f} us")
