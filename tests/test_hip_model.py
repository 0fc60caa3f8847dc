"""GPU parity tests, model level: the HIP path (through the reference-shaped modules) vs the CPU
oracle on the same seeded inputs, and vs the committed goldens produced by the real reference.

Tolerances (north star): logits within 1e-3 (f32 path); argmax masks bit-exact wherever the
oracle's own top-2 margin exceeds that tolerance (a margin below the logit tolerance cannot be
decided bit-exactly by ANY two f32 implementations with different summation order; the golden's
minimum margin is 3e-6).
"""

import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

gdlhip = pytest.importorskip("gdlhip")
import oracle  # noqa: E402
from gdlhip import nn as gnn  # noqa: E402
from geo_deep_learning.models.encoders.dofa_v2 import DOFAv2  # noqa: E402
from geo_deep_learning.models.segmentation.dofa import DOFASegmentationModel  # noqa: E402
from oracle.model import dice_loss_multiclass  # noqa: E402
from oracle import procedural_state_dict, synthetic_batch  # noqa: E402

DEV = "cuda"
LOGIT_TOL = 1e-3


def _sub(t, sc, sp, off=1):
    return t.detach().float().cpu()[:, ::sc, off::sp, off::sp].numpy()


def _drop_masks(depth, rate, batch, seed):
    g = np.random.default_rng([seed, depth, batch, 7])
    dpr = torch.linspace(0, rate, depth).tolist()
    masks = []
    for i in range(depth):
        pair = []
        for j in range(2):
            m = (g.uniform(size=batch) < 1.0 - dpr[i]).astype(np.float32)
            if i == depth - 1 and j == 0:
                m[0] = 0.0
            if i == depth // 2 and j == 1:
                m[-1] = 0.0
            pair.append(torch.from_numpy(m))
        masks.append(tuple(pair))
    return masks


def _aux_mask(batch, ch, seed):
    g = np.random.default_rng([seed, batch, ch, 11])
    return torch.from_numpy((g.uniform(size=(batch, ch)) < 0.9).astype(np.float32))


def _grad_sample(g, n):
    f = g.detach().float().cpu().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy()


TIGHT = ("head.", "aux_head.", "decoder.fpn_bottleneck.")


def _grad_check(name, grad, ref_norm, ref_sample, n):
    """End-to-end gradient parity.  Layers next to the loss must agree tightly (1e-3).  Deeper
    layers see ReLU-mask flips caused by f32 forward round-off (|pre-activation| below the 1e-4
    forward error): the ORACLE ITSELF differs by up to 5e-3 between torch-on-CPU and torch-on-GPU
    on exactly these layers (tools/debug_grads.py, DESIGN.md "gradient tolerance"), so they are
    held to 2e-2; each kernel is held to 1e-4 on its own in test_hip_ops.py.  Conv biases that
    feed a train-mode BN have an analytically zero gradient (reference: rounding noise)."""
    got = grad.double().norm().item()
    if name.endswith("conv.bias") and name.startswith("neck."):
        assert got <= 1e-6 and ref_norm <= 1e-6, (name, got, ref_norm)
        return
    rel = 1e-3 if name.startswith(TIGHT) else 2e-2
    assert abs(got - ref_norm) <= rel * ref_norm + 2e-5, (name, got, ref_norm)
    # element check, robust to an isolated flipped ReLU mask (with 32 pixels per channel in the tiny
    # aux head ONE flip changes a whole output channel's row): >= 99 % of sampled elements agree
    smp = _grad_sample(grad, n)
    bad = np.abs(smp - ref_sample) > 5 * rel * np.abs(ref_sample).max() + 1e-9
    assert bad.mean() <= 0.01, (name, float(bad.mean()))


# pinned counts of argmax mismatches (observed on MI355X, round 4, `pytest -s`: 0 of 262144 / 0 of 1048576 pixels, although 0 / 127
# pixels have a top-2 margin below the logit tolerance): the masks ARE bit-exact
MULTIBAND_MAX_MISMATCH = {"dofa_base": 0, "dofa_large": 0}


def _mask_check(got_mask, ref_logits, ref_mask, max_mismatch=None):
    """Bit-exact wherever the reference's top-2 margin exceeds the logit tolerance; the mismatches that remain (pixels
    whose two best logits are closer than any two f32 summation orders can resolve) are COUNTED, printed (pytest -s / the
    failure message) and bounded: at most 1e-4 of the pixels, or ``max_mismatch`` pixels when the caller pins a count."""
    top2 = ref_logits.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    decided = margin > LOGIT_TOL
    got = got_mask.cpu().numpy()
    diff = got != ref_mask
    n_bad, n_pix, n_undecided = int(diff.sum()), diff.size, int((~decided).sum())
    worst = float(margin[diff].max()) if n_bad else 0.0
    print(f"mask check: {n_bad} of {n_pix} pixels differ ({n_undecided} pixels have a top-2 margin <= {LOGIT_TOL:g}; "
          f"largest margin among the differing pixels {worst:.2e})")
    assert not (diff & decided).any(), f"argmax differs at a pixel whose margin {worst:.2e} exceeds the tolerance {LOGIT_TOL:g}"
    bound = max_mismatch if max_mismatch is not None else int(1e-4 * n_pix)
    assert n_bad <= bound, f"{n_bad} mismatching pixels of {n_pix} (bound {bound}, {n_undecided} undecidable)"


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = np.load(golden_dir / "dofa_tiny.npz")
    meta = json.loads(str(g["meta"]))
    ref = oracle.DOFASegmentationModel("dofa_tiny_test", (meta["img"],) * 2, num_classes=meta["num_classes"],
                                       _encoder_kwargs=meta["tiny"], freeze_layers=["encoder"])
    sd = procedural_state_dict(ref, meta["seed"])
    ref.load_state_dict(sd)
    enc = DOFAv2(img_size=meta["img"], pretrained=False, **meta["tiny"])
    model = DOFASegmentationModel(enc, (meta["img"],) * 2, num_classes=meta["num_classes"], pretrained=False,
                                  freeze_layers=["encoder"])
    assert list(model.state_dict().keys()) == list(ref.state_dict().keys())
    model.load_state_dict(sd)
    model = model.to(DEV)
    batch = synthetic_batch(meta["batch"], 3, meta["img"], meta["num_classes"], meta["seed"])
    return g, meta, ref, model, batch


def test_tiny_dynamic_kernel(tiny):
    g, meta, ref, model, batch = tiny
    with torch.no_grad():
        w, b = model.encoder.patch_embed.generate(batch["wavelengths"])
    np.testing.assert_allclose(w.cpu().numpy(), g["dyn_weight"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(b.cpu().numpy(), g["dyn_bias"], atol=2e-4, rtol=0)


def test_tiny_eval_f32(tiny):
    g, meta, ref, model, batch = tiny
    model.eval()
    x = batch["image"].to(DEV)
    with torch.no_grad():
        taps = model.encoder(x, batch["wavelengths"])
        feats = model.neck(taps)
        r = model(x, batch["wavelengths"])
    for i in range(4):
        np.testing.assert_allclose(taps[i].cpu().numpy(), g[f"eval_tap{i}"], atol=5e-4, rtol=0, err_msg=f"tap{i}")
        np.testing.assert_allclose(feats[i].float().cpu().numpy(), g[f"eval_neck{i}"], atol=5e-4, rtol=0,
                                   err_msg=f"neck{i}")
    np.testing.assert_allclose(r.out.cpu().numpy(), g["eval_out"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(r.aux.cpu().numpy(), g["eval_aux"], atol=LOGIT_TOL, rtol=0)
    assert r.out.dtype == torch.float32 and r.out.is_contiguous() and r.out.shape == (2, 5, 112, 112)
    _mask_check(gnn.predict_mask(r.out), torch.from_numpy(g["eval_out"]), g["eval_mask"])


def test_tiny_eval_bf16_autocast(tiny):
    g, meta, ref, model, batch = tiny
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        r = model(batch["image"].to(DEV), batch["wavelengths"])
    ref_out = torch.from_numpy(g["eval_out"])
    err = (r.out.cpu() - ref_out).abs().max().item()
    assert err < 0.05 * ref_out.abs().max().item(), f"bf16 logits err {err}"
    agree = (gnn.predict_mask(r.out).cpu().numpy() == g["eval_mask"]).mean()
    assert agree > 0.97, agree


def test_tiny_train_step_f32(tiny):
    g, meta, ref, model, batch = tiny
    model.train()
    for p in model.parameters():
        p.grad = None
    b = meta["batch"]
    # fresh running stats (the eval tests do not touch them, but keep the test order-independent)
    model.load_state_dict(procedural_state_dict(ref, meta["seed"]))
    masks = _drop_masks(meta["tiny"]["depth"], 0.1, b, meta["seed"])
    r = model(batch["image"].to(DEV), batch["wavelengths"], masks, _aux_mask(b, 256, meta["seed"]))
    np.testing.assert_allclose(r.out.detach().cpu().numpy(), g["train_out"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(r.aux.detach().cpu().numpy(), g["train_aux"], atol=LOGIT_TOL, rtol=0)
    y = batch["mask"].squeeze(1).long().to(DEV)
    crit = gnn.DiceLoss(mode="multiclass")
    loss = crit(r.out, y) + 0.4 * crit(r.aux, y)
    assert abs(loss.item() - float(g["train_loss"])) < 1e-5
    loss.backward()
    params = dict(model.named_parameters())
    got_names = sorted(n for n, p in params.items() if p.grad is not None)
    assert got_names == sorted(meta["grad_names"])
    for n in meta["grad_names"]:
        _grad_check(n, params[n].grad, float(g["gradnorm/" + n]), g["grad/" + n], 2048)
    bufs = dict(model.named_buffers())
    for k in g.files:
        if k.startswith("buf/"):
            np.testing.assert_allclose(bufs[k[4:]].cpu().numpy(), g[k], atol=2e-5, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("freeze", [["encoder.patch_embed"], None], ids=["generator_frozen", "all_trainable"])
def test_tiny_train_vit_blocks_unfrozen(tiny, freeze):
    """Unfrozen encoder: the ViT blocks + cls_token train through the HIP block backward (attention / LayerNorm /
    LayerScale / GELU kernels) and, with nothing frozen, the dynamic weight generator through its own autograd
    node (FCRes, the post-norm transformer layer, fc_weight / fc_bias).  Checked against the CPU oracle."""
    g, meta, _, _, batch = tiny
    nc, img, b, seed = meta["num_classes"], meta["img"], meta["batch"], meta["seed"]
    ref = oracle.DOFASegmentationModel("dofa_tiny_test", (img,) * 2, num_classes=nc, _encoder_kwargs=meta["tiny"],
                                       freeze_layers=freeze).train()
    sd = procedural_state_dict(ref, seed)
    ref.load_state_dict(sd)
    enc = DOFAv2(img_size=img, pretrained=False, **meta["tiny"])
    model = DOFASegmentationModel(enc, (img,) * 2, num_classes=nc, pretrained=False, freeze_layers=freeze)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    masks = _drop_masks(meta["tiny"]["depth"], 0.1, b, seed)
    am = _aux_mask(b, 256, seed)
    y = batch["mask"].squeeze(1).long()
    ro = ref(batch["image"], batch["wavelengths"], masks, am)
    lo = dice_loss_multiclass(ro.out, y) + 0.4 * dice_loss_multiclass(ro.aux, y)
    lo.backward()
    r = model(batch["image"].to(DEV), batch["wavelengths"], masks, am)
    crit = gnn.DiceLoss(mode="multiclass")
    loss = crit(r.out, y.to(DEV)) + 0.4 * crit(r.aux, y.to(DEV))
    loss.backward()
    assert abs(loss.item() - lo.item()) < 1e-5
    refp = dict(ref.named_parameters())
    n_enc = 0
    for n, p in model.named_parameters():
        rg = refp[n].grad
        assert (p.grad is None) == (rg is None), n
        if rg is None or (n.endswith("conv.bias") and n.startswith("neck.")):
            continue
        n_enc += n.startswith("encoder.")
        # absolute floor: the last block's fc2.bias shifts every token of a tap by a per-channel constant,
        # which the neck's train-mode BN removes -> analytically zero gradient (rounding noise in both)
        err, ref_n = (p.grad.cpu() - rg).norm().item(), rg.norm().item()
        assert err <= 3e-2 * ref_n + 2e-6, (n, err, ref_n)
    assert n_enc > (60 if freeze is None else 40)
    if freeze is None:
        # bf16 autocast: the generator still runs in f32 and its gradients stay finite and aligned with the f32
        # reference.  Checked at batch 8: with the fixture's B = 2 the PSP branch's 1x1 bins put TWO samples through a
        # batch-statistics BatchNorm, which is a sign function of their difference -- any rounding change flips
        # channels and the encoder gradients turn by tens of degrees (an input perturbation of 1e-6 already gives
        # cos 0.96, tools/debug/blk_probe.py), so bf16-vs-f32 alignment is not a meaningful quantity there.
        b8 = 8
        batch8 = synthetic_batch(b8, 3, img, nc, seed + 1)
        masks8 = _drop_masks(meta["tiny"]["depth"], 0.1, b8, seed)
        am8 = _aux_mask(b8, 256, seed)
        y8 = batch8["mask"].squeeze(1).long()
        for p in list(model.parameters()) + list(ref.parameters()):
            p.grad = None
        ro = ref(batch8["image"], batch8["wavelengths"], masks8, am8)
        (dice_loss_multiclass(ro.out, y8) + 0.4 * dice_loss_multiclass(ro.aux, y8)).backward()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            r = model(batch8["image"].to(DEV), batch8["wavelengths"], masks8, am8)
            lb = crit(r.out, y8.to(DEV)) + 0.4 * crit(r.aux, y8.to(DEV))
        lb.backward()
        coss = {}
        for n, p in model.named_parameters():
            rg = refp[n].grad
            if n.startswith("encoder.") and rg is not None and rg.numel() >= 64 and rg.norm() > 1e-7:
                assert torch.isfinite(p.grad).all(), n
                gq = p.grad.float().cpu()
                coss[n] = float((gq * rg).sum() / (gq.norm() * rg.norm() + 1e-30))
        worst = sorted(coss.items(), key=lambda kv: kv[1])[:5]
        assert coss["encoder.patch_embed.weight_generator.fc_weight.weight"] > 0.9, worst
        assert np.median(list(coss.values())) > 0.95, worst


def test_tiny_train_bf16_runs_and_descends(tiny):
    """bf16 autocast training: loss close to the f32 reference and Adam steps reduce it."""
    g, meta, ref, model, batch = tiny
    model.load_state_dict(procedural_state_dict(ref, meta["seed"]))
    model.train()
    opt = gnn.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
    crit = gnn.DiceLoss(mode="multiclass")
    x, y = batch["image"].to(DEV), batch["mask"].squeeze(1).long().to(DEV)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            r = model(x, batch["wavelengths"])
            loss = crit(r.out, y) + 0.4 * crit(r.aux, y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert abs(losses[0] - float(g["train_loss"])) < 0.05
    assert losses[-1] < losses[0], losses


def test_frozen_contract(tiny):
    """freeze_layers substring match (base.py:40-44); mixed wavelengths in a batch are rejected (dofa_v2.py:437-442)."""
    g, meta, ref, model, batch = tiny
    assert all(not p.requires_grad for n, p in model.named_parameters() if "encoder" in n)
    assert all(p.requires_grad for n, p in model.named_parameters() if "encoder" not in n)
    with pytest.raises(ValueError):
        wv = torch.tensor([[0.6, 0.5, 0.4], [0.6, 0.5, 0.41]])
        with torch.no_grad():
            model.encoder(batch["image"].to(DEV), wv)


@pytest.fixture(scope="module")
def base_model():
    ref = oracle.DOFASegmentationModel("dofa_base", (512, 512), num_classes=5, freeze_layers=["encoder"])
    sd = procedural_state_dict(ref, 42)
    ref.load_state_dict(sd)
    model = DOFASegmentationModel("dofa_base", (512, 512), num_classes=5, pretrained=False,
                                  freeze_layers=["encoder"])
    model.load_state_dict(sd)
    return ref, model.to(DEV), sd


def test_base_512_eval_f32(base_model, golden_dir):
    """BASELINE config 2 at full size: vs the golden (real reference) AND vs the oracle run here."""
    ref, model, _ = base_model
    g = np.load(golden_dir / "dofa_base_512_eval.npz")
    meta = json.loads(str(g["meta"]))
    batch = synthetic_batch(meta["batch"], 3, 512, meta["num_classes"], meta["seed"])
    model.eval()
    ref.eval()
    x = batch["image"].to(DEV)
    with torch.no_grad():
        taps = model.encoder(x, batch["wavelengths"])
        r = model(x, batch["wavelengths"])
        o = ref(batch["image"], batch["wavelengths"])
    for i in range(4):
        np.testing.assert_allclose(_sub(taps[i], 16, 5), g[f"tap{i}_s"], atol=5e-4, rtol=0, err_msg=f"tap{i}")
    np.testing.assert_allclose(_sub(r.out, 1, 8, 3), g["out_s8"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(_sub(r.aux, 1, 8, 3), g["aux_s8"], atol=LOGIT_TOL, rtol=0)
    assert (r.out.cpu() - o.out).abs().max().item() < LOGIT_TOL
    assert (r.aux.cpu() - o.aux).abs().max().item() < LOGIT_TOL
    # observed (round 4, `pytest -s`): 0 of 524288 pixels differ from the real reference's mask, although 166 pixels have a top-2
    # margin below the logit tolerance -- pinned: the full-size mask is bit-exact, not merely "exact where decidable"
    _mask_check(gnn.predict_mask(r.out), o.out, g["mask"], max_mismatch=0)


def test_base_512_eval_bf16(base_model, golden_dir):
    ref, model, _ = base_model
    g = np.load(golden_dir / "dofa_base_512_eval.npz")
    meta = json.loads(str(g["meta"]))
    batch = synthetic_batch(meta["batch"], 3, 512, meta["num_classes"], meta["seed"])
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        r = model(batch["image"].to(DEV), batch["wavelengths"])
    got = _sub(r.out, 1, 8, 3)
    scale = np.abs(g["out_s8"]).max()
    err = np.abs(got - g["out_s8"]).max()
    assert err < 0.06 * scale, f"bf16 logits err {err} vs scale {scale}"
    agree = (gnn.predict_mask(r.out).cpu().numpy() == g["mask"]).mean()
    assert agree > 0.97, agree


@pytest.mark.parametrize("seed", [77, 78, 79])
def test_base_512_eval_bf16_no_worse_than_torch_autocast(base_model, seed):
    """Round-4 review, parity item 2 (round 6: three seeds): the benchmarked dtype was only held to "within 6 % of the f32 logits".  Here it is pinned to
    what the reference itself does in that dtype: the SAME oracle model run by torch under `autocast("cuda", bfloat16)` on this GPU
    (f32 residual stream, bf16 GEMM operands, f32 accumulation: the reference's `precision: bf16-mixed` path).  Both are
    approximations of the f32 CPU oracle; the build's error against it must not exceed torch's own by more than a quarter (max and
    RMS over all 5 x 512 x 512 logits of two tiles), and the two bf16 masks must agree with the f32 mask about equally often."""
    import copy
    ref, model, _ = base_model
    batch = synthetic_batch(2, 3, 512, 5, seed)
    model.eval()
    ref.eval()
    with torch.no_grad():
        o32 = ref(batch["image"], batch["wavelengths"]).out
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ours = model(batch["image"].to(DEV), batch["wavelengths"]).out.float().cpu()
        tref = copy.deepcopy(ref).to(DEV).eval()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            theirs = tref(batch["image"].to(DEV), batch["wavelengths"].to(DEV)).out.float().cpu()
        del tref
    e_ours, e_torch = (ours - o32), (theirs - o32)
    mx = (e_ours.abs().max().item(), e_torch.abs().max().item())
    rms = (e_ours.pow(2).mean().sqrt().item(), e_torch.pow(2).mean().sqrt().item())
    want = o32.softmax(1).argmax(1)
    agree = ((ours.softmax(1).argmax(1) == want).float().mean().item(), (theirs.softmax(1).argmax(1) == want).float().mean().item())
    print(f"bf16 vs the f32 oracle -- build: max {mx[0]:.3e} rms {rms[0]:.3e} mask agreement {agree[0]:.5f}; "
          f"torch autocast: max {mx[1]:.3e} rms {rms[1]:.3e} mask agreement {agree[1]:.5f}")
    assert mx[0] <= 1.25 * mx[1] + 1e-3 and rms[0] <= 1.25 * rms[1] + 1e-4, (mx, rms)
    assert agree[0] >= agree[1] - 5e-3, agree


def _grad_errors(named_got, named_ref):
    """relative L2 error per parameter (parameters with a non-trivial reference gradient only)"""
    out = {}
    for n, g in named_got.items():
        r = named_ref[n]
        if r.norm().item() > 1e-8 and r.numel() >= 16:
            out[n] = ((g.double() - r.double()).norm() / r.double().norm()).item()
    return out


def test_base_512_train_bf16_no_worse_than_torch_autocast(base_model):
    """The training twin of the test above (round-5 review, item 9): ONE bf16 training step of configs[1] -- DOFA-base + UperNet at
    512 x 512, encoder frozen, DropPath / Dropout2d draws pinned -- through the HIP path under autocast, and the same step of the
    oracle run by torch under `autocast("cuda", bfloat16)` (the reference's `precision: bf16-mixed`); truth = the oracle's f32
    step on this GPU.  Loss error and the per-parameter gradient errors (relative L2, all 60-odd trainable tensors of neck,
    decoder and heads) of the build must not exceed torch's own by more than a quarter."""
    import copy
    ref, model, _ = base_model
    b, depth = 2, 12
    batch = synthetic_batch(b, 3, 512, 5, 91)
    g = np.random.default_rng(91)
    masks = [(torch.from_numpy((g.uniform(size=b) < 0.9).astype(np.float32)), torch.from_numpy((g.uniform(size=b) < 0.9).astype(np.float32)))
             for _ in range(depth)]
    aux = torch.from_numpy((g.uniform(size=(b, 256)) < 0.9).astype(np.float32))
    tgt = batch["mask"].squeeze(1).long()
    tref = copy.deepcopy(ref).to(DEV).train()
    runs = {}
    for name, amp in (("f32", False), ("torch_bf16", True)):
        tref.zero_grad(set_to_none=True)
        # (MIOpen's train-mode batch_norm segfaults on the 1 x 1 pyramid-pooling maps in this image: torch's native kernels)
        with torch.backends.cudnn.flags(enabled=False), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            o = tref(batch["image"].to(DEV), batch["wavelengths"].to(DEV), [(a.to(DEV), c.to(DEV)) for a, c in masks], aux.to(DEV))
            loss = oracle.model.training_loss(o, batch["mask"].to(DEV))
        with torch.backends.cudnn.flags(enabled=False):
            loss.backward()
        runs[name] = (loss.item(), {n: p.grad.detach().float().cpu() for n, p in tref.named_parameters() if p.grad is not None})
    del tref
    model.train()
    model.zero_grad(set_to_none=True)
    crit = gnn.DiceLoss()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(batch["image"].to(DEV), batch["wavelengths"], masks, aux)
        loss = crit(o.out, tgt.to(DEV)) + 0.4 * crit(o.aux, tgt.to(DEV))
    loss.backward()
    ours = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    l32, g32 = runs["f32"]
    lt, gt = runs["torch_bf16"]
    assert set(ours) == set(g32), set(ours) ^ set(g32)
    e_ours, e_torch = _grad_errors(ours, g32), _grad_errors(gt, g32)
    med = lambda d: float(np.median(list(d.values())))      # noqa: E731
    print(f"bf16 training step vs the f32 oracle -- loss error: build {abs(loss.item() - l32):.2e}, torch autocast {abs(lt - l32):.2e}; "
          f"gradient error (relative L2 over {len(e_ours)} tensors): build median {med(e_ours):.4f} max {max(e_ours.values()):.4f}, "
          f"torch autocast median {med(e_torch):.4f} max {max(e_torch.values()):.4f}")
    assert abs(loss.item() - l32) <= 1.25 * abs(lt - l32) + 2e-3
    assert med(e_ours) <= 1.25 * med(e_torch) + 1e-3
    assert max(e_ours.values()) <= 1.25 * max(e_torch.values()) + 1e-2


def test_base_512_train_f32(base_model, golden_dir):
    ref, model, sd = base_model
    g = np.load(golden_dir / "dofa_base_512_train.npz")
    meta = json.loads(str(g["meta"]))
    b = meta["batch"]
    batch = synthetic_batch(b, 3, 512, meta["num_classes"], meta["seed"])
    model.load_state_dict(sd)
    model.train()
    for p in model.parameters():
        p.grad = None
    r = model(batch["image"].to(DEV), batch["wavelengths"], _drop_masks(12, 0.1, b, meta["seed"]),
              _aux_mask(b, 256, meta["seed"]))
    np.testing.assert_allclose(_sub(r.out, 1, 8, 3), g["out_s8"], atol=LOGIT_TOL, rtol=0)
    y = batch["mask"].squeeze(1).long().to(DEV)
    crit = gnn.DiceLoss(mode="multiclass")
    lm, la = crit(r.out, y), crit(r.aux, y)
    loss = lm + 0.4 * la
    assert abs(lm.item() - float(g["loss_main"])) < 1e-5 and abs(la.item() - float(g["loss_aux"])) < 1e-5
    loss.backward()
    params = dict(model.named_parameters())
    for n in meta["grad_names"]:
        _grad_check(n, params[n].grad, float(g["gradnorm/" + n]), g["grad/" + n], 1024)
    bufs = dict(model.named_buffers())
    for k in g.files:
        if k.startswith("buf/"):
            np.testing.assert_allclose(bufs[k[4:]].cpu().numpy(), g[k], atol=2e-5, rtol=1e-4, err_msg=k)


def test_size_independent_properties(base_model):
    """Properties that hold at any size: batch independence in eval, determinism, mask range."""
    ref, model, _ = base_model
    model.eval()
    batch = synthetic_batch(3, 3, 512, 5, 123)
    x = batch["image"].to(DEV)
    with torch.no_grad():
        full = model(x, batch["wavelengths"]).out
        again = model(x, batch["wavelengths"]).out
        single = model(x[1:2], batch["wavelengths"]).out
    assert torch.equal(full, again), "forward is not deterministic"
    assert (full[1:2] - single).abs().max().item() < 1e-4, "eval output depends on batch composition"
    m = gnn.predict_mask(full)
    assert m.dtype == torch.int64 and m.min().item() >= 0 and m.max().item() < 5


def test_script_model_wrapper(tiny):
    """tools/script_model.py contract: raw 0-255 tile -> normalise -> model -> softmax probabilities."""
    from geo_deep_learning.tools.script_model import SegmentationScriptModel
    from oracle.model import RGB_MEAN, RGB_STD
    g, meta, ref, model, batch = tiny
    sd = procedural_state_dict(ref, meta["seed"])       # earlier tests trained the shared model: fresh weights
    ref.load_state_dict(sd)
    model.load_state_dict(sd)
    ref.eval()
    wrap = SegmentationScriptModel(model, wavelengths=batch["wavelengths"], device=torch.device(DEV),
                                   num_classes=meta["num_classes"], input_shape=(1, 3, meta["img"], meta["img"]),
                                   mean=RGB_MEAN, std=RGB_STD)
    probs = wrap(batch["image_u8"])
    with torch.no_grad():
        want = ref(batch["image"], batch["wavelengths"]).out.softmax(dim=1)
    assert probs.shape == want.shape
    assert (probs.cpu() - want).abs().max().item() < 2e-4
    assert (probs.sum(1) - 1).abs().max().item() < 1e-5


@pytest.mark.parametrize("encoder,size,bands,tol", [("dofa_base", 512, 6, 1e-3), ("dofa_large", 1024, 10, 2e-3)],
                         ids=["config3_base_6band_512", "config4_large_10band_1024"])
def test_multiband_configs_match_oracle(encoder, size, bands, tol):
    """BASELINE configs[3] / configs[4] shapes (6-band DOFA-base 512^2; 10-band DOFA-large 1024^2), one tile:
    f32 logits vs the CPU oracle, masks exact where the oracle's margin exceeds the tolerance, bf16 agreement."""
    seed = 21
    ref = oracle.DOFASegmentationModel(encoder, (size, size), num_classes=5, freeze_layers=["encoder"]).eval()
    sd = procedural_state_dict(ref, seed)
    ref.load_state_dict(sd)
    model = DOFASegmentationModel(encoder, (size, size), num_classes=5, pretrained=False, freeze_layers=["encoder"])
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    batch = synthetic_batch(1, bands, size, 5, seed)
    x = batch["image"].to(DEV)
    warned = set(gnn._WARNED_FALLBACK)
    with torch.no_grad():
        yo = ref(batch["image"], batch["wavelengths"]).out
        y = model(x, batch["wavelengths"]).out
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = model(x, batch["wavelengths"]).out
    # round 5: DOFA-large's 292 / 146 / 73 / 36 pyramid (x 8.11 for the top level) no longer sends `fpn_bottleneck` to its unfused path
    assert set(gnn._WARNED_FALLBACK) == warned, set(gnn._WARNED_FALLBACK) - warned
    scale = max(1.0, yo.abs().max().item())
    assert (y.cpu() - yo).abs().max().item() < tol * scale
    top2 = yo.topk(2, dim=1).values
    decided = ((top2[:, 0] - top2[:, 1]) > 2 * tol * scale).numpy()
    want = yo.softmax(1).argmax(1).numpy()
    got = gnn.predict_mask(y).cpu().numpy()
    n_bad = int((got != want).sum())
    print(f"mask check ({encoder}, {bands} bands, {size}^2): {n_bad} of {want.size} pixels differ, {int((~decided).sum())} pixels "
          f"have a top-2 margin <= {2 * tol * scale:g}; max logit error {(y.cpu() - yo).abs().max().item():.2e}")
    assert (got == want)[decided].all()
    assert n_bad <= MULTIBAND_MAX_MISMATCH[encoder], n_bad
    assert (yb.float().cpu() - yo).abs().max().item() < 0.08 * yo.abs().max().item()
    assert (gnn.predict_mask(yb).cpu().numpy() == want).mean() > 0.95


def test_ddp_syncbn_wrapper_single_rank(tiny):
    """The N>1 launch path of bench.py (SyncBatchNorm conversion + DistributedDataParallel with
    gradient_as_bucket_view + the fused optimizer) on a 1-rank RCCL group: same loss and gradients as the bare model,
    and the optimizer step works on the bucket-view gradients."""
    import os
    import torch.distributed as dist
    g, meta, ref, _, batch = tiny
    nc, img, b, seed = meta["num_classes"], meta["img"], meta["batch"], meta["seed"]
    sd = procedural_state_dict(ref, seed)

    def build():
        enc = DOFAv2(img_size=img, pretrained=False, **meta["tiny"])
        m = DOFASegmentationModel(enc, (img,) * 2, num_classes=nc, pretrained=False, freeze_layers=["encoder"])
        m.load_state_dict(sd)
        return m.to(DEV).train()
    masks = _drop_masks(meta["tiny"]["depth"], 0.1, b, seed)
    am = _aux_mask(b, 256, seed)
    y = batch["mask"].squeeze(1).long().to(DEV)
    crit = gnn.DiceLoss(mode="multiclass")

    def step(model):
        r = model(batch["image"].to(DEV), batch["wavelengths"], masks, am)
        loss = crit(r.out, y) + 0.4 * crit(r.aux, y)
        loss.backward()
        return loss.item()
    bare = build()
    l0 = step(bare)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(build())
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], gradient_as_bucket_view=True,
                                                        find_unused_parameters=False)
        l1 = step(ddp)
        assert abs(l1 - l0) < 1e-6
        for (n, p), (_, q) in zip(bare.named_parameters(), m.named_parameters()):
            if p.grad is None:
                assert q.grad is None, n
                continue
            assert torch.allclose(p.grad, q.grad, atol=1e-6, rtol=1e-5), n
        before = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
        opt = gnn.FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
        opt.step()
        moved = [n for n, p in m.named_parameters() if p.requires_grad and not torch.equal(before[n], p.detach())]
        expect = [n for n, p in m.named_parameters() if p.requires_grad and p.grad.abs().max() > 0]
        assert moved == expect and len(moved) > 60      # zero-gradient biases in front of a BN do not move
        opt.zero_grad(set_to_none=True)
        l2 = step(ddp)                      # a second step through the same buckets
        assert abs(l2 - l1) > 0 and l2 == l2
    finally:
        dist.destroy_process_group()
