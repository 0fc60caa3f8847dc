"""GPU parity tests for the SegFormer path: HIP modules vs goldens produced by
the real reference and vs the CPU oracle.  Forward / inference only this round."""

import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

gdlhip = pytest.importorskip("gdlhip")
from gdlhip import nn as gnn  # noqa: E402
from geo_deep_learning.models.segmentation.segformer import SegFormerSegmentationModel  # noqa: E402
from oracle import procedural_state_dict, synthetic_batch  # noqa: E402
from oracle.segformer import SegFormerSegmentationModel as OracleSegFormer  # noqa: E402

DEV = "cuda"


def _sub(t, sc, sp, off=1):
    return t.detach().float().cpu()[:, ::sc, off::sp, off::sp].numpy()


def _build(enc, seed):
    ora = OracleSegFormer(enc, 3, 5).eval()
    sd = procedural_state_dict(ora, seed)
    ora.load_state_dict(sd)
    m = SegFormerSegmentationModel(enc, 3, None, None, 5)
    m.load_state_dict(sd)
    return ora, m.to(DEV).eval()


def test_segformer_b1_small_f32(golden_dir):
    g = np.load(golden_dir / "segformer.npz")
    seed = json.loads(str(g["meta"]))["seed"]
    ora, m = _build("mit_b1", seed)
    batch = synthetic_batch(2, 3, 64, 5, seed)
    with torch.no_grad():
        feats = m.encoder(batch["image"].to(DEV))
        y = m(batch["image"].to(DEV))
    for i, f in enumerate(feats):
        np.testing.assert_allclose(f.float().cpu().numpy(), g[f"b1_feat{i}"], atol=5e-4, rtol=0, err_msg=f"feat{i}")
    np.testing.assert_allclose(y.cpu().numpy(), g["b1_out"], atol=1e-3, rtol=0)


def test_segformer_b2_512(golden_dir):
    """BASELINE config 3 (SegFormer-B2, 512x512): f32 parity at full size + bf16 agreement."""
    g = np.load(golden_dir / "segformer.npz")
    seed = json.loads(str(g["meta"]))["seed"]
    ora, m = _build("mit_b2", seed)
    batch = synthetic_batch(1, 3, 512, 5, seed)
    x = batch["image"].to(DEV)
    with torch.no_grad():
        feats = m.encoder(x)
        y = m(x)
        yo = ora(batch["image"])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = m(x)
    for i, f in enumerate(feats):
        np.testing.assert_allclose(_sub(f, 8, 3), g[f"b2_feat{i}_s"], atol=5e-4, rtol=0, err_msg=f"feat{i}")
    np.testing.assert_allclose(_sub(y, 1, 8, 3), g["b2_out_s8"], atol=1e-3, rtol=0)
    assert (y.cpu() - yo).abs().max().item() < 1e-3
    mask = gnn.predict_mask(y).cpu().numpy()
    top2 = yo.topk(2, dim=1).values
    decided = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
    assert (mask == g["b2_mask"])[decided].all()
    assert (yb.cpu() - yo).abs().max().item() < 0.08 * yo.abs().max().item()
    assert (gnn.predict_mask(yb).cpu().numpy() == g["b2_mask"]).mean() > 0.97


def test_segformer_training_fails_loudly():
    m = SegFormerSegmentationModel("mit_b1", 3, None, None, 5).to(DEV).train()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 3, 64, 64, device=DEV))
