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


def test_segformer_b2_512_train_bf16_no_worse_than_torch_autocast():
    """configs[2] at the benchmarked dtype (round-5 review, item 9): one bf16 training step of SegFormer-B2 at 512 x 512 (every
    parameter trainable, DropPath / Dropout2d draws pinned) through the HIP path under autocast against the oracle run by torch
    under `autocast("cuda", bfloat16)`; truth = the oracle's f32 step on this GPU.  Loss error and per-parameter gradient errors
    (relative L2) of the build at most a quarter above torch's own."""
    import copy
    seed, nc, b = 11, 5, 2
    ora, m = _build("mit_b2", seed)
    m.train()
    batch = synthetic_batch(b, 3, 512, nc, seed)
    masks = mit_drop_masks(m.encoder.depths, 0.1, b, seed)
    dmask = chan_mask(b, m.decoder.linear_pred.in_channels, seed)
    tref = copy.deepcopy(ora).to(DEV).train()
    tgt = batch["mask"].squeeze(1).long().to(DEV)
    runs = {}
    for name, amp in (("f32", False), ("torch_bf16", True)):
        tref.zero_grad(set_to_none=True)
        # (MIOpen's train-mode batch_norm segfaults on the 1 x 1 pyramid-pooling maps in this image: torch's native kernels)
        with torch.backends.cudnn.flags(enabled=False), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            yo = tref(batch["image"].to(DEV), [(a.to(DEV), c.to(DEV)) for a, c in masks], dmask.to(DEV))
            lo = dice_loss_multiclass(yo.float(), tgt)
        with torch.backends.cudnn.flags(enabled=False):
            lo.backward()
        runs[name] = (lo.item(), {n: p.grad.detach().float().cpu() for n, p in tref.named_parameters() if p.grad is not None})
    del tref
    _, loss = _train_step(m, batch, masks, dmask, autocast=True)
    ours = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None}
    l32, g32 = runs["f32"]
    lt, gt = runs["torch_bf16"]

    def errs(got):
        out = {}
        for n, g in got.items():
            r = g32[n]
            if r.norm().item() > 1e-8 and r.numel() >= 16:
                out[n] = ((g.double() - r.double()).norm() / r.double().norm()).item()
        return out
    e_ours, e_torch = errs(ours), errs(gt)
    med = lambda d: float(np.median(list(d.values())))      # noqa: E731
    print(f"SegFormer-B2 bf16 training step vs the f32 oracle -- loss error: build {abs(loss.item() - l32):.2e}, torch autocast "
          f"{abs(lt - l32):.2e}; gradient error over {len(e_ours)} tensors: build median {med(e_ours):.4f} max {max(e_ours.values()):.4f}, "
          f"torch autocast median {med(e_torch):.4f} max {max(e_torch.values()):.4f}")
    assert abs(loss.item() - l32) <= 1.25 * abs(lt - l32) + 2e-3
    assert med(e_ours) <= 1.25 * med(e_torch) + 1e-3
    assert max(e_ours.values()) <= 1.25 * max(e_torch.values()) + 1e-2


# ------------------------------------------------------------------ S7: channel-adaptive stem (use_dynamic_encoder=True)
def _build_dynamic(enc, seed, nc=5):
    ora = OracleSegFormer(enc, 3, nc, use_dynamic_encoder=True).eval()
    sd = procedural_state_dict(ora, seed)
    ora.load_state_dict(sd)
    m = SegFormerSegmentationModel(enc, 3, None, None, nc, use_dynamic_encoder=True)
    assert sorted(m.state_dict().keys()) == sorted(sd.keys())
    m.load_state_dict(sd)
    return ora, m.to(DEV).eval()


@pytest.mark.parametrize("E,C,P", [(64, 6, 50), (32, 3, 33), (64, 16, 7), (32, 1, 9)])
def test_band_pooling_kernels_match_autograd(E, C, P):
    """gdl_chan_weights_* / gdl_chan_pool_* against the oracle's DynamicChannelEmbed math under torch autograd."""
    from gdlhip import tnn
    from oracle.segformer import DynamicChannelEmbed
    torch.manual_seed(3)
    B = 2
    ora = DynamicChannelEmbed(7, 4, E, 128)
    for p in ora.parameters():
        p.data.mul_(3.0)                                  # spread the band logits
    conv = torch.randn(B, C, P, E)
    dagg = torch.randn(B, P, E)
    k = ora.band_codes(C)
    # oracle math on the conv tensor (spatial conv skipped: it is the stem GEMM tested elsewhere)
    cr = conv.clone().requires_grad_(True)
    v = cr * ora.weight_gen(k)[None, :, None, :]
    a0, a2 = ora.channel_attention[0], ora.channel_attention[2]
    feat = torch.cat([v, k[None, :, None, :].expand(B, C, P, -1)], dim=-1)
    s = (feat @ a0.weight[:, :, 0].t() + a0.bias).relu() @ a2.weight[0, :, 0] + a2.bias
    ref = (v * s.softmax(dim=1)[..., None]).sum(dim=1)
    ref.backward(dagg)
    import copy
    dev = copy.deepcopy(ora).to(DEV)
    for p in dev.parameters():
        p.grad = None
    cd = conv.to(DEV).requires_grad_(True)
    got = tnn.chan_pool(cd, k.to(DEV), dev.weight_gen, dev.channel_attention)
    assert (got.cpu() - ref.detach()).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    got.backward(dagg.to(DEV))
    assert (cd.grad.cpu() - cr.grad).abs().max().item() < 1e-4 * max(1.0, cr.grad.abs().max().item())
    for (n, pr), (_, pd) in zip(ora.named_parameters(), dev.named_parameters()):
        if n == "channel_attention.2.bias":      # shifts every band's logit alike: analytically zero gradient
            assert pd.grad.abs().max().item() == 0.0 and pr.grad.abs().max().item() < 1e-4
        elif n.startswith(("weight_gen", "channel_attention")):
            scale = max(pr.grad.abs().max().item(), 1e-4)
            assert (pd.grad.cpu() - pr.grad).abs().max().item() < 2e-4 * scale + 1e-6, n


def test_dynamic_segformer_f32(golden_dir):
    """Stem tokens (6 and 3 bands), eval logits and the full train step vs the real reference's outputs."""
    g = np.load(golden_dir / "segformer_dynamic.npz")
    meta = json.loads(str(g["meta"]))
    seed, b, nc, bands = meta["seed"], meta["batch"], meta["num_classes"], meta["bands"]
    ora, m = _build_dynamic(meta["encoder"], seed, nc)
    batch = synthetic_batch(b, bands, meta["size"], nc, seed)
    x = batch["image"].to(DEV)
    with torch.no_grad():
        tok, h, w = m.encoder.dynamic_patch_embed1(x)
        tok3, _, _ = m.encoder.dynamic_patch_embed1(x[:, :3].contiguous())
        y = m(x)
    assert (h, w) == (meta["size"] // 4, meta["size"] // 4)
    np.testing.assert_allclose(tok.cpu().numpy(), g["stem_tokens"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(tok3.cpu().numpy(), g["stem_tokens_3band"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(y.cpu().numpy(), g["eval_out"], atol=1e-3, rtol=0)
    m.train()
    masks = mit_drop_masks(meta["depths"], 0.1, b, seed)
    dmask = chan_mask(b, m.decoder.linear_pred.in_channels, seed)
    out, loss = _train_step(m, batch, masks, dmask)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["train_out"], atol=1e-3, rtol=0)
    assert abs(loss.item() - float(g["train_loss"])) < 1e-4
    named = [(n, p.grad) for n, p in m.named_parameters()]
    assert all(gr is not None for _, gr in named)
    assert sorted(n for n, _ in named) == sorted(meta["grad_names"])
    worst = check_grads(named, g, tol=3e-2, tight=("decoder.linear_pred.", "encoder.dynamic_patch_embed1.proj."),
                        tight_tol=5e-3)
    print("dynamic segformer train f32: worst 99%-quantile relative grad error", worst)


def test_dynamic_segformer_bf16_train_step(golden_dir):
    g = np.load(golden_dir / "segformer_dynamic.npz")
    meta = json.loads(str(g["meta"]))
    seed, b, nc, bands = meta["seed"], meta["batch"], meta["num_classes"], meta["bands"]
    ora, m = _build_dynamic(meta["encoder"], seed, nc)
    m.train()
    batch = synthetic_batch(b, bands, meta["size"], nc, seed)
    masks = mit_drop_masks(meta["depths"], 0.1, b, seed)
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
    assert np.median(dots) > 0.97, np.median(dots)


def test_segformer_b0_reference_config():
    """configs/segformer_config_RGB.yaml:43 ships encoder mit_b0 (embed dims 32/64/160/256, 32-wide heads): none of its
    widths fills a bf16 K chunk, so this exercises the channel-tail kernels.  f32 logits vs the CPU oracle; the bf16
    train step stays close to the f32 oracle's loss and gradients."""
    seed, nc, b, size = 11, 5, 2, 128
    ora, m = _build("mit_b0", seed)
    batch = synthetic_batch(b, 3, size, nc, seed)
    x = batch["image"].to(DEV)
    with torch.no_grad():
        y = m(x)
        yo = ora(batch["image"])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = m(x)
    assert (y.cpu() - yo).abs().max().item() < 1e-3
    assert (yb.cpu() - yo).abs().max().item() < 0.08 * yo.abs().max().item()
    ora.train()
    m.train()
    masks = mit_drop_masks(m.encoder.depths, 0.1, b, seed)
    dmask = chan_mask(b, m.decoder.linear_pred.in_channels, seed)
    out, loss = _train_step(m, batch, masks, dmask, autocast=True)
    yo = ora(batch["image"], masks, dmask)
    lo = dice_loss_multiclass(yo, batch["mask"].squeeze(1).long())
    lo.backward()
    assert abs(loss.item() - lo.item()) < 2e-2
    ref = dict(ora.named_parameters())
    dots = []
    for n, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), n
        r = ref[n].grad.flatten()
        if r.norm().item() > 1e-6 and r.numel() >= 64:
            gq = p.grad.flatten().float().cpu()
            dots.append(float(torch.dot(gq, r) / (gq.norm() * r.norm() + 1e-30)))
    assert np.median(dots) > 0.97, np.median(dots)
