"""GPU parity tests for the SegFormer path: HIP modules vs goldens produced by
the real reference and vs the CPU oracle: inference and the full training step (every parameter
trainable, DropPath / Dropout2d masks pinned)."""

import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

gdlhip = pytest.importorskip("gdlhip")
from gdlhip import nn as gnn  # noqa: E402
from geo_deep_learning.models.segmentation.segformer import SegFormerSegmentationModel  # noqa: E402
from _recipes import chan_mask, check_grads, mit_drop_masks  # noqa: E402
from oracle import procedural_state_dict, synthetic_batch  # noqa: E402
from oracle.model import dice_loss_multiclass  # noqa: E402
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


def _train_step(m, batch, masks, dmask, autocast=False):
    x, y = batch["image"].to(DEV), batch["mask"].to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = m(x, masks, dmask)
        loss = gnn.DiceLoss()(out, y.squeeze(1).long())
    loss.backward()
    return out, loss


def test_segformer_train_step_f32(golden_dir):
    """S1-S6 training: logits, loss, EVERY parameter gradient and the BN running stats vs the reference."""
    g = np.load(golden_dir / "segformer_train.npz")
    meta = json.loads(str(g["meta"]))
    seed, b, nc = meta["seed"], meta["batch"], meta["num_classes"]
    ora, m = _build(meta["encoder"], seed)
    m.train()
    batch = synthetic_batch(b, 3, meta["size"], nc, seed)
    masks = mit_drop_masks(m.encoder.depths, 0.1, b, seed)
    dmask = chan_mask(b, m.decoder.linear_pred.in_channels, seed)
    out, loss = _train_step(m, batch, masks, dmask)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["train_out"], atol=1e-3, rtol=0)
    assert abs(loss.item() - float(g["train_loss"])) < 1e-4
    named = [(n, p.grad) for n, p in m.named_parameters()]
    assert all(gr is not None for _, gr in named)
    assert sorted(n for n, _ in named) == sorted(meta["grad_names"])
    worst = check_grads(named, g, tol=2e-2, tight=("decoder.linear_pred.",), tight_tol=2e-3)
    print("segformer train f32: worst 99%-quantile relative grad error", worst)
    for n, buf in m.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(buf.cpu().numpy(), g["buf/" + n], atol=1e-4, rtol=1e-4)


def test_segformer_train_step_bf16_and_optimizer(golden_dir):
    """bf16 autocast train step: loss close to the reference's, gradients finite and aligned with the f32
    reference (cosine), fused optimizer step changes every parameter."""
    g = np.load(golden_dir / "segformer_train.npz")
    meta = json.loads(str(g["meta"]))
    seed, b, nc = meta["seed"], meta["batch"], meta["num_classes"]
    ora, m = _build(meta["encoder"], seed)
    m.train()
    batch = synthetic_batch(b, 3, meta["size"], nc, seed)
    masks = mit_drop_masks(m.encoder.depths, 0.1, b, seed)
    dmask = chan_mask(b, m.decoder.linear_pred.in_channels, seed)
    out, loss = _train_step(m, batch, masks, dmask, autocast=True)
    assert abs(loss.item() - float(g["train_loss"])) < 2e-2
    from _recipes import grad_sample
    dots = []
    for n, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), n
        ref = g["grad/" + n]
        got = grad_sample(p.grad, 512)
        if np.linalg.norm(ref) > 1e-6 and ref.size >= 64:
            dots.append(float(np.dot(got, ref) / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30)))
    assert np.median(dots) > 0.98, np.median(dots)
    assert np.quantile(dots, 0.05) > 0.85, np.quantile(dots, 0.05)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = gnn.FusedAdam(m.parameters(), lr=1e-3, weight_decay=0.01)
    opt.step()
    changed = sum(int(not torch.equal(before[n], p.detach())) for n, p in m.named_parameters())
    assert changed == len(before)


def test_segformer_b2_512_train_matches_oracle():
    """BASELINE config 3 at full size (B=1): HIP f32 train step vs the CPU oracle on the same inputs."""
    seed, nc = 7, 5
    ora, m = _build("mit_b2", seed)
    ora.train()
    m.train()
    batch = synthetic_batch(1, 3, 512, nc, seed)
    masks = mit_drop_masks(m.encoder.depths, 0.1, 1, seed)
    masks = [(torch.ones(1), torch.ones(1)) for _ in masks]      # B=1: keep every path (a dropped sample kills BN)
    dmask = chan_mask(1, m.decoder.linear_pred.in_channels, seed)
    out, loss = _train_step(m, batch, masks, dmask)
    yo = ora(batch["image"], masks, dmask)
    lo = dice_loss_multiclass(yo, batch["mask"].squeeze(1).long())
    lo.backward()
    assert (out.detach().cpu() - yo.detach()).abs().max().item() < 2e-3
    assert abs(loss.item() - lo.item()) < 1e-4
    ref = dict(ora.named_parameters())
    bad = []
    for n, p in m.named_parameters():
        r = ref[n].grad
        # absolute floor: biases feeding the train-mode BN of linear_fuse (linear_c*.proj.bias, norm4.bias)
        # have an analytically zero gradient -- rounding noise in both implementations
        err, rn = (p.grad.cpu() - r).norm().item(), r.norm().item()
        if err > 3e-2 * rn + 2e-6:
            bad.append((n, err, rn))
    assert not bad, bad[:10]
