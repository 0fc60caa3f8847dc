#!/usr/bin/env python
"""GPU diagnostic: per-parameter gradient error of the HIP training step vs the CPU oracle."""

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))

import oracle  # noqa: E402
from gdlhip import nn as gnn  # noqa: E402
from geo_deep_learning.models.encoders.dofa_v2 import DOFAv2  # noqa: E402
from geo_deep_learning.models.segmentation.dofa import DOFASegmentationModel  # noqa: E402


def main(cfg="tiny", autocast=False):
    if cfg == "tiny":
        tiny = dict(patch_size=14, embed_dim=128, depth=4, num_heads=2, out_indices=[0, 1, 2, 3])
        img, b = 112, 2
        ref = oracle.DOFASegmentationModel("dofa_tiny_test", (img, img), num_classes=5, _encoder_kwargs=tiny,
                                           freeze_layers=["encoder"])
        enc = DOFAv2(img_size=img, pretrained=False, **tiny)
        model = DOFASegmentationModel(enc, (img, img), num_classes=5, pretrained=False, freeze_layers=["encoder"])
        depth = 4
    else:
        img, b, depth = 512, 2, 12
        ref = oracle.DOFASegmentationModel("dofa_base", (img, img), num_classes=5, freeze_layers=["encoder"])
        model = DOFASegmentationModel("dofa_base", (img, img), num_classes=5, pretrained=False,
                                      freeze_layers=["encoder"])
    sd = oracle.procedural_state_dict(ref, 42)
    ref.load_state_dict(sd)
    model.load_state_dict(sd)
    model = model.cuda().train()
    ref.train()
    batch = oracle.synthetic_batch(b, 3, img, 5, 42)
    masks = [(torch.ones(b), torch.ones(b))] * depth
    am = (torch.rand(b, 256, generator=torch.Generator().manual_seed(1)) < 0.9).float()
    r = ref(batch["image"], batch["wavelengths"], masks, am)
    lr = oracle.model.training_loss(r, batch["mask"])
    lr.backward()
    y = batch["mask"].squeeze(1).long().cuda()
    crit = gnn.DiceLoss()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        o = model(batch["image"].cuda(), batch["wavelengths"], masks, am)
        lo = crit(o.out, y) + 0.4 * crit(o.aux, y)
    lo.backward()
    print(f"loss hip {lo.item():.7f} oracle {lr.item():.7f}  logits err {(o.out.cpu() - r.out).abs().max().item():.2e}")
    rp = dict(ref.named_parameters())
    # calibration: the SAME oracle module run by torch on the GPU (rocBLAS/MIOpen summation order)
    import copy
    ref_gpu = copy.deepcopy(ref).cuda()
    for p in ref_gpu.parameters():
        p.grad = None
    rg = ref_gpu(batch["image"].cuda(), batch["wavelengths"].cuda(), [(a.cuda(), c.cuda()) for a, c in masks],
                 am.cuda())
    lg = oracle.model.training_loss(rg, batch["mask"].cuda())
    lg.backward()
    print(f"[calibration] oracle-on-GPU loss {lg.item():.7f} logits err vs oracle-CPU "
          f"{(rg.out.cpu() - r.out).abs().max().item():.2e}")
    gp = dict(ref_gpu.named_parameters())
    for n, p in ref.named_parameters():
        if p.grad is not None and n.endswith("weight"):
            rel = (gp[n].grad.cpu() - p.grad).norm().item() / (p.grad.norm().item() + 1e-12)
            print(f"[calibration] {n:48s} oracle GPU-vs-CPU rel {rel:.2e}")
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g, gr = p.grad.float().cpu(), rp[n].grad
        err = (g - gr).abs()
        rel = (g - gr).norm().item() / (gr.norm().item() + 1e-12)
        flag = " <<<" if rel > 2e-3 else ""
        print(f"{n:48s} |g| {gr.norm().item():.3e} rel {rel:.2e} maxabs {err.max().item():.2e}{flag}")
        if rel > 2e-3 and g.dim() == 4:
            e = err.flatten()
            top = e.topk(5).indices
            for i in top.tolist():
                idx = torch.unravel_index(torch.tensor(i), g.shape)
                print("      idx", [int(v) for v in idx], "hip", g.flatten()[i].item(), "ref", gr.flatten()[i].item())
            per_n = err.amax(dim=(1, 2, 3))
            bad = (per_n > 10 * err.median()).nonzero().flatten().tolist()
            print("      bad out-channels:", bad[:20], "count", len(bad))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tiny", autocast=len(sys.argv) > 2)
