#!/usr/bin/env python
"""Diagnostic: UNet++ / ResNet18 train-step gradients of the HIP model vs the CPU oracle (f32), for image sizes and dtypes given
as `size:dtype` arguments (default 128:f32 128:bf16 512:f32 512:bf16); per-parameter relative L2 errors in module order."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
sys.path.insert(0, str(ROOT))
from gdlhip import nn as gnn  # noqa: E402
from geo_deep_learning.models.segmentation.unetplusplus import UnetPlusPlus  # noqa: E402
from oracle import procedural_state_dict, synthetic_batch  # noqa: E402
from oracle.model import dice_loss_multiclass  # noqa: E402
from oracle.unetpp import UnetPlusPlus as OracleUnetPlusPlus  # noqa: E402

cases = sys.argv[1:] or ["128:f32", "128:bf16", "512:f32", "512:bf16"]
verbose = "-v" in cases
cases = [c for c in cases if c != "-v"]
for case in cases:
    size, dt = case.split(":")
    size = int(size)
    ora = OracleUnetPlusPlus("resnet18", 3, 5)
    sd = procedural_state_dict(ora, 21)
    ora.load_state_dict(sd)
    m = UnetPlusPlus("resnet18", encoder_weights=None, classes=5)
    m.load_state_dict(sd)
    m = m.cuda()
    batch = synthetic_batch(2, 3, size, 5, 21)
    ora.train(); m.train()
    yo = ora(batch["image"])
    lo = dice_loss_multiclass(yo, batch["mask"].squeeze(1).long())
    lo.backward()
    x, yt = batch["image"].cuda(), batch["mask"].cuda()
    if dt == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
            loss = gnn.DiceLoss()(y, yt)
    else:
        y = m(x)
        loss = gnn.DiceLoss()(y, yt)
    loss.backward()
    ref = dict(ora.named_parameters())
    rels = []
    for n, p in m.named_parameters():
        g, r = p.grad.float().cpu(), ref[n].grad
        rels.append((float((g - r).norm() / (r.norm() + 1e-30)), n, float(r.norm()), float(g.norm())))
    srt = sorted(rels)
    print(f"== {case}: logits rel err {(y.float().cpu() - yo).abs().max().item() / yo.abs().max().item():.3e}, loss {loss.item():.6f} vs {lo.item():.6f}, "
          f"grad rel L2 median {srt[len(srt) // 2][0]:.4f}, worst {srt[-1][0]:.4f} ({srt[-1][1]})", flush=True)
    if verbose:
        for r in rels:
            print(f"   {r[0]:8.4f}  |ref| {r[2]:.3e} |got| {r[3]:.3e}  {r[1]}")
