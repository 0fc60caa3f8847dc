#!/usr/bin/env python
"""Debug probe: cosine between bf16-autocast gradients of the tiny DOFA model (everything trainable) and the f32
CPU oracle, per parameter group, under different kernel-selection hooks."""
import ctypes
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402
from oracle import procedural_state_dict, synthetic_batch  # noqa: E402
from oracle.model import dice_loss_multiclass  # noqa: E402
from gdlhip import _lib, nn as gnn  # noqa: E402
from geo_deep_learning.models.encoders.dofa_v2 import DOFAv2  # noqa: E402
from geo_deep_learning.models.segmentation.dofa import DOFASegmentationModel  # noqa: E402
from test_hip_model import _aux_mask, _drop_masks  # noqa: E402

DEV = "cuda"
g = np.load(ROOT / "tests/golden/dofa_tiny.npz")
meta = json.loads(str(g["meta"]))
nc, img, b, seed = meta["num_classes"], meta["img"], meta["batch"], meta["seed"]
ref = oracle.DOFASegmentationModel("dofa_tiny_test", (img,) * 2, num_classes=nc, _encoder_kwargs=meta["tiny"],
                                   freeze_layers=None).train()
sd = procedural_state_dict(ref, seed)
ref.load_state_dict(sd)
batch = synthetic_batch(b, 3, img, nc, seed)
masks = _drop_masks(meta["tiny"]["depth"], 0.1, b, seed)
am = _aux_mask(b, 256, seed)
y = batch["mask"].squeeze(1).long()
ro = ref(batch["image"], batch["wavelengths"], masks, am)
lo = dice_loss_multiclass(ro.out, y) + 0.4 * dice_loss_multiclass(ro.aux, y)
lo.backward()
refp = dict(ref.named_parameters())
import os
if os.environ.get('GDL_LIB'):
    _lib.LIB_PATH = Path(os.environ['GDL_LIB'])
lib = _lib.load()
for fn in ("gdl_debug_force_wgrad_small", "gdl_debug_force_conv_variant", "gdl_debug_set_conv_tap_packing"):
    getattr(lib, fn).argtypes = [ctypes.c_int]


def run(tag, bwd_inside=False):
    enc = DOFAv2(img_size=img, pretrained=False, **meta["tiny"])
    model = DOFASegmentationModel(enc, (img,) * 2, num_classes=nc, pretrained=False, freeze_layers=None)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    crit = gnn.DiceLoss(mode="multiclass")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        print("compute dtype inside autocast:", gnn.compute_dtype())
        r = model(batch["image"].to(DEV), batch["wavelengths"], masks, am)
        lb = crit(r.out, y.to(DEV)) + 0.4 * crit(r.aux, y.to(DEV))
        if bwd_inside:
            lb.backward()
    if not bwd_inside:
        lb.backward()
    bad, good = [], 0
    for n, p in model.named_parameters():
        rg = refp[n].grad
        if rg is None or rg.norm() < 1e-6 or rg.numel() < 64:
            continue
        gq = p.grad.float().cpu()
        cos = float((gq * rg).sum() / (gq.norm() * rg.norm() + 1e-30))
        if cos < 0.9:
            bad.append((n.replace("encoder.", "e.").replace("patch_embed.weight_generator.", "gen."), round(cos, 2)))
        else:
            good += 1
    print(tag, "loss", round(lb.item(), 5), "ref", round(lo.item(), 5), "good", good, "bad", len(bad), flush=True)
    print("   ", [b for b in bad if "gen." not in b[0]][:40], flush=True)


run("default")
