"""Debug probe: token-stream values and gradients at every ViT block boundary of the tiny DOFA model under bf16
autocast, saved to gpurun_out/blk_<TAG>.npz.  NOISE=<rel> perturbs the input image; GDL_LIB=<path> loads another
build of libgdlhip.so.  Used to show that with B = 2 (two-sample BatchNorm in the PSP 1x1 bins) the encoder gradients
are chaotic: NOISE=1e-6 already turns them by 15 degrees."""
import json, os, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle
from oracle import procedural_state_dict, synthetic_batch
from gdlhip import _lib, nn as gnn, tnn
if os.environ.get('GDL_LIB'):
    _lib.LIB_PATH = Path(os.environ['GDL_LIB'])
from geo_deep_learning.models.encoders.dofa_v2 import DOFAv2
from geo_deep_learning.models.segmentation.dofa import DOFASegmentationModel
from test_hip_model import _aux_mask, _drop_masks
DEV = "cuda"
g = np.load(ROOT / "tests/golden/dofa_tiny.npz"); meta = json.loads(str(g["meta"]))
nc, img, b, seed = meta["num_classes"], meta["img"], meta["batch"], meta["seed"]
ref = oracle.DOFASegmentationModel("dofa_tiny_test", (img,) * 2, num_classes=nc, _encoder_kwargs=meta["tiny"], freeze_layers=None)
sd = procedural_state_dict(ref, seed)
batch = synthetic_batch(b, 3, img, nc, seed)
masks = _drop_masks(meta["tiny"]["depth"], 0.1, b, seed); am = _aux_mask(b, 256, seed)
y = batch["mask"].squeeze(1).long()
enc = DOFAv2(img_size=img, pretrained=False, **meta["tiny"])
model = DOFASegmentationModel(enc, (img,) * 2, num_classes=nc, pretrained=False, freeze_layers=None)
model.load_state_dict(sd); model = model.to(DEV).train()
keep = []
orig = tnn.vit_block
def wrapped(x, *a, **k):
    x.retain_grad(); keep.append(("in", x))
    out = orig(x, *a, **k)
    out.retain_grad(); keep.append(("out", out))
    return out
tnn.vit_block = wrapped
crit = gnn.DiceLoss(mode="multiclass")
with torch.autocast("cuda", dtype=torch.bfloat16):
    im = batch["image"].to(DEV)
    if os.environ.get("NOISE"):
        torch.manual_seed(1)
        im = im * (1 + float(os.environ["NOISE"]) * torch.randn_like(im))
    r = model(im, batch["wavelengths"], masks, am)
    lb = crit(r.out, y.to(DEV)) + 0.4 * crit(r.aux, y.to(DEV))
lb.backward()
out = {}
for i, (kind, t) in enumerate(keep):
    out[f"{i:02d}_{kind}"] = t.detach().float().cpu().numpy()
    out[f"{i:02d}_{kind}_grad"] = t.grad.detach().float().cpu().numpy()
for n, p in model.named_parameters():
    if n.startswith("encoder.blocks.3."):
        out["p/" + n] = p.grad.float().cpu().numpy()
np.savez(ROOT / "gpurun_out" / f"blk_{os.environ.get('TAG','cur')}.npz", **out)
print("saved", len(out), lb.item())
