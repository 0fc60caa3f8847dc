"""Pin the SegFormer oracle against outputs of the REAL reference (tests/golden/segformer.npz,
produced by tools/make_goldens.py --only segformer)."""

import json

import numpy as np
import torch

from _recipes import chan_mask, check_grads, mit_drop_masks
from oracle import procedural_state_dict, synthetic_batch
from oracle.model import dice_loss_multiclass
from oracle.segformer import MIT_VARIANTS, SegFormerSegmentationModel


def _sub(t, sc, sp, off=1):
    return t.detach()[:, ::sc, off::sp, off::sp].numpy()


def test_segformer_oracle_matches_reference(golden_dir):
    g = np.load(golden_dir / "segformer.npz")
    seed = json.loads(str(g["meta"]))["seed"]
    m = SegFormerSegmentationModel("mit_b1", 3, 5).eval()
    m.load_state_dict(procedural_state_dict(m, seed))
    batch = synthetic_batch(2, 3, 64, 5, seed)
    with torch.no_grad():
        feats = m.encoder(batch["image"])
        y = m(batch["image"])
    for i, f in enumerate(feats):
        np.testing.assert_allclose(f.numpy(), g[f"b1_feat{i}"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(y.numpy(), g["b1_out"], atol=1e-4, rtol=0)

    m = SegFormerSegmentationModel("mit_b2", 3, 5).eval()
    m.load_state_dict(procedural_state_dict(m, seed))
    batch = synthetic_batch(1, 3, 512, 5, seed)
    with torch.no_grad():
        feats = m.encoder(batch["image"])
        y = m(batch["image"])
    for i, f in enumerate(feats):
        np.testing.assert_allclose(_sub(f, 8, 3), g[f"b2_feat{i}_s"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(_sub(y, 1, 8, 3), g["b2_out_s8"], atol=2e-4, rtol=0)
    mask = y.softmax(1).argmax(1).numpy()
    top2 = y.topk(2, dim=1).values
    decided = ((top2[:, 0] - top2[:, 1]) > 1e-3).numpy()
    assert (mask == g["b2_mask"])[decided].all()


def test_segformer_oracle_train_step_matches_reference(golden_dir):
    """Full train step (DropPath / Dropout2d masks pinned, BN batch statistics) vs the real reference."""
    g = np.load(golden_dir / "segformer_train.npz")
    meta = json.loads(str(g["meta"]))
    seed, b, nc = meta["seed"], meta["batch"], meta["num_classes"]
    m = SegFormerSegmentationModel(meta["encoder"], 3, nc).train()
    m.load_state_dict(procedural_state_dict(m, seed))
    batch = synthetic_batch(b, 3, meta["size"], nc, seed)
    depths = MIT_VARIANTS[meta["encoder"]]["depths"]
    y = m(batch["image"], mit_drop_masks(depths, 0.1, b, seed), chan_mask(b, m.decoder.linear_pred.in_channels, seed))
    np.testing.assert_allclose(y.detach().numpy(), g["train_out"], atol=2e-4, rtol=0)
    loss = dice_loss_multiclass(y, batch["mask"].squeeze(1).long())
    assert abs(loss.item() - float(g["train_loss"])) < 1e-5
    loss.backward()
    named = [(n, p.grad) for n, p in m.named_parameters()]
    assert sorted(n for n, _ in named) == sorted(meta["grad_names"])
    check_grads(named, g, tol=2e-3)
    for n, buf in m.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(buf.numpy(), g["buf/" + n], atol=1e-5, rtol=1e-5)


def test_dynamic_segformer_oracle_matches_reference(golden_dir):
    """S7 (use_dynamic_encoder=True): channel-adaptive stem on 6 and 3 bands, eval logits, and the full train step
    (every gradient) vs the real reference."""
    g = np.load(golden_dir / "segformer_dynamic.npz")
    meta = json.loads(str(g["meta"]))
    seed, b, nc, bands = meta["seed"], meta["batch"], meta["num_classes"], meta["bands"]
    m = SegFormerSegmentationModel(meta["encoder"], 3, nc, use_dynamic_encoder=True).eval()
    m.load_state_dict(procedural_state_dict(m, seed))
    batch = synthetic_batch(b, bands, meta["size"], nc, seed)
    with torch.no_grad():
        tok, _, _ = m.encoder.dynamic_patch_embed1(batch["image"])
        tok3, _, _ = m.encoder.dynamic_patch_embed1(batch["image"][:, :3])
        y = m(batch["image"])
    np.testing.assert_allclose(tok.numpy(), g["stem_tokens"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(tok3.numpy(), g["stem_tokens_3band"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(y.numpy(), g["eval_out"], atol=2e-4, rtol=0)
    m.train()
    y = m(batch["image"], mit_drop_masks(meta["depths"], 0.1, b, seed), chan_mask(b, m.decoder.linear_pred.in_channels, seed))
    np.testing.assert_allclose(y.detach().numpy(), g["train_out"], atol=2e-4, rtol=0)
    loss = dice_loss_multiclass(y, batch["mask"].squeeze(1).long())
    assert abs(loss.item() - float(g["train_loss"])) < 1e-5
    loss.backward()
    named = [(n, p.grad) for n, p in m.named_parameters()]
    assert sorted(n for n, _ in named) == sorted(meta["grad_names"])
    check_grads(named, g, tol=6e-3)      # 64^2 tiles leave stage 4 with 2x2 tokens: f32 summation-order noise
