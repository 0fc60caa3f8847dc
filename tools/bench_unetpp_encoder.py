#!/usr/bin/env python
"""Train / inference rate of UNet++ with a given encoder (default: the reference's shipped resnext101_32x8d) at 512^2, bf16:
tools/bench_unetpp_encoder.py [encoder] [batch].  For the record only -- the grouped 3x3 convolutions run as block-diagonal
dense filters (gdlhip.cnn.mark_groups), BASELINE.json's UNet++ configuration is ResNet18."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import nn as gnn  # noqa: E402
from geo_deep_learning.models.segmentation.unetplusplus import UnetPlusPlus  # noqa: E402

enc = sys.argv[1] if len(sys.argv) > 1 else "resnext101_32x8d"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
m = UnetPlusPlus(enc, encoder_weights=None, classes=5).cuda()
opt = gnn.FusedAdam(m.parameters(), lr=1e-4, max_grad_norm=1.0)
x = torch.randn(B, 3, 512, 512, device="cuda")
y = torch.randint(0, 5, (B, 1, 512, 512), device="cuda")
loss_fn = gnn.DiceLoss()


def train():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = loss_fn(m(x), y)
    loss.backward()
    opt.step()
    return loss


def infer():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return gnn.predict_mask(m(x))


for name, fn, mode in (("train", train, True), ("inference", infer, False)):
    m.train(mode)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"UNet++ / {enc}, batch {B}, 512^2 bf16: {name} {dt * 1e3:.1f} ms/step = {B / dt:.1f} tiles/s"
          + (f", loss {float(out):.4f}" if mode else ""), flush=True)
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, parameters {sum(p.numel() for p in m.parameters()) / 1e6:.1f} M")
