"""GPU parity tests for the UNet++ path: the new HIP kernels vs plain PyTorch, and the
HIP model (eval + full train step, f32 and bf16) vs the CPU oracle on the same seeded inputs.  The oracle for
this row is "parity unpinned" (smp / torchvision are not available; oracle/unetpp.py header)."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

gdlhip = pytest.importorskip("gdlhip")
from gdlhip import cnn, ops  # noqa: E402
from gdlhip import nn as gnn  # noqa: E402
from geo_deep_learning.models.segmentation.unetplusplus import UnetPlusPlus  # noqa: E402
from oracle import procedural_state_dict, synthetic_batch  # noqa: E402
from oracle.model import dice_loss_multiclass  # noqa: E402
from oracle.unetpp import UnetPlusPlus as OracleUnetPlusPlus  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (1, 9, 7, 8), (2, 32, 24, 128)])
def test_maxpool(dtype, B, H, W, C):
    x = rnd(B, H, W, C).to(dtype).float()
    x[0, :6, :6] = x[0, :6, :6].clamp_min(0).round()          # ties (post-ReLU zeros): first maximum takes the gradient
    xr = x.clone().requires_grad_()
    y = F.max_pool2d(xr.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    dy = rnd(*y.shape, seed=1).to(dtype).float()
    y.backward(dy)
    xd = x.to(DEV, dtype)
    yd = ops.maxpool3x3s2(xd)
    assert torch.equal(yd.float().cpu(), y.detach())
    dx = ops.maxpool3x3s2_bwd(xd, dy.to(DEV, dtype))
    tol = 0 if dtype == torch.float32 else 2e-2 * dy.abs().max().item()
    assert (dx.float().cpu() - xr.grad).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_upcat_and_add_relu(dtype):
    x, s1, s2 = rnd(2, 8, 6, 64).to(dtype).float(), rnd(2, 16, 12, 128, seed=1).to(dtype).float(), \
        rnd(2, 16, 12, 64, seed=2).to(dtype).float()
    ref_in = [t.clone().requires_grad_() for t in (x, s1, s2)]
    up = F.interpolate(ref_in[0].permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    ref = torch.cat([up, ref_in[1], ref_in[2]], dim=-1)
    g = rnd(*ref.shape, seed=3).to(dtype).float()
    ref.backward(g)
    dev_in = [t.to(DEV, dtype).requires_grad_() for t in (x, s1, s2)]
    out = cnn.up_cat(dev_in[0], dev_in[1:])
    assert torch.equal(out.float().cpu(), ref.detach())
    out.backward(g.to(DEV, dtype))
    for d, r in zip(dev_in, ref_in):
        assert (d.grad.float().cpu() - r.grad).abs().max().item() <= (0 if dtype == torch.float32 else 1e-2 * r.grad.abs().max().item())
    a, b = rnd(2, 5, 5, 64, seed=4).to(dtype).float(), rnd(2, 5, 5, 64, seed=5).to(dtype).float()
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    F.relu(ar + br).backward(g[:, :5, :5, :64].contiguous())
    ad, bd = a.to(DEV, dtype).requires_grad_(), b.to(DEV, dtype).requires_grad_()
    o = cnn.add_relu(ad, bd)
    o.backward(g[:, :5, :5, :64].contiguous().to(DEV, dtype))
    assert (o.float().cpu() - F.relu(a + b)).abs().max().item() <= (0 if dtype == torch.float32 else 2e-2)
    assert torch.equal(ad.grad.float().cpu(), ar.grad.to(dtype).float()) and torch.equal(ad.grad, bd.grad)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,stride,k", [(64, 32, 1, 3), (64, 128, 2, 3), (64, 128, 2, 1), (128, 16, 1, 3),
                                               (32, 16, 1, 3), (16, 16, 1, 3), (64, 20, 1, 3), (24, 40, 2, 3)])
def test_conv_bn_node_padded(dtype, cin, cout, stride, k):
    """conv-BN-ReLU training node with narrow channels (K-chunk tails), output padding to the 16-byte granularity
    and stride vs torch autograd."""
    B, H = 2, 12
    # (the parameters come from torch's global generator, seeded per test id by conftest.py: unseeded, about one in four bf16
    #  instances of the 6 x 6-output case flipped a ReLU mask at a near-zero pre-activation and missed the tolerance)
    conv = torch.nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
    bn = torch.nn.BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(conv.weight.to(dtype).float())
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
    x = rnd(B, H, H, cin).to(dtype).float()
    xr = x.clone().requires_grad_()
    y = F.relu(bn(conv(xr.permute(0, 3, 1, 2)))).permute(0, 2, 3, 1)
    gy = rnd(*y.shape, seed=2).to(dtype).float()
    y.backward(gy)
    cg = torch.nn.Conv2d(cin, cout, k, stride, k // 2, bias=False).to(DEV).to(memory_format=torch.channels_last)
    bg = torch.nn.BatchNorm2d(cout).to(DEV)
    cg.load_state_dict(conv.state_dict())
    bg.load_state_dict({k_: v for k_, v in bn.state_dict().items()})
    bg.running_mean.zero_(); bg.running_var.fill_(1.0); bg.num_batches_tracked.zero_()
    xd = x.to(DEV, dtype).requires_grad_()
    yd = cnn.conv_bn(xd, cg.weight, bg, stride=stride, pad=k // 2)
    npad = cnn.pad_to(cout, cnn.grain(dtype))
    assert yd.shape[-1] == npad
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    sc = y.abs().max().item()
    assert (yd[..., :cout].float().cpu() - y.detach()).abs().max().item() <= tol * sc
    assert yd[..., cout:].abs().max().item() == 0 if npad > cout else True
    gpad = torch.zeros((*gy.shape[:3], npad))
    gpad[..., :cout] = gy
    yd.backward(gpad.to(DEV, dtype))
    for got, ref, nm in ((xd.grad, xr.grad, "dx"), (cg.weight.grad, conv.weight.grad, "dw"),
                         (bg.weight.grad, bn.weight.grad, "dgamma"), (bg.bias.grad, bn.bias.grad, "dbeta")):
        err = (got.float().cpu() - ref).abs().max().item()
        assert err <= 3 * tol * ref.abs().max().item(), (nm, err, ref.abs().max().item())
    assert torch.allclose(bg.running_mean.cpu(), bn.running_mean, atol=1e-2 if dtype == torch.bfloat16 else 1e-5)
    assert torch.allclose(bg.running_var.cpu(), bn.running_var, atol=1e-2 if dtype == torch.bfloat16 else 1e-5)


def _build(enc, seed, classes=5):
    ora = OracleUnetPlusPlus(enc, 3, classes)
    sd = procedural_state_dict(ora, seed)
    ora.load_state_dict(sd)
    m = UnetPlusPlus(enc, encoder_weights=None, classes=classes)
    assert list(m.state_dict().keys()) == list(ora.state_dict().keys())
    m.load_state_dict(sd)
    return ora, m.to(DEV)


@pytest.mark.parametrize("enc", ["resnet18", "resnet34"])
def test_unetpp_eval_f32_and_bf16(enc):
    ora, m = _build(enc, 11)
    ora.eval(); m.eval()
    batch = synthetic_batch(2, 3, 128, 5, 11)
    x = batch["image"].to(DEV)
    with torch.no_grad():
        yo = ora(batch["image"])
        y = m(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = m(x)
    assert y.shape == yo.shape == (2, 5, 128, 128)
    assert (y.cpu() - yo).abs().max().item() < 1e-3 * max(1.0, yo.abs().max().item())
    assert (yb.cpu() - yo).abs().max().item() < 0.06 * yo.abs().max().item()
    mask = gnn.predict_mask(y).cpu()
    top2 = yo.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 2e-3
    assert (mask == yo.softmax(1).argmax(1))[decided].all()


def test_unetpp_512_eval_matches_oracle():
    """BASELINE config 1 shape (UNet++ ResNet18, 512x512 RGB)."""
    ora, m = _build("resnet18", 3)
    ora.eval(); m.eval()
    batch = synthetic_batch(1, 3, 512, 5, 3)
    with torch.no_grad():
        yo = ora(batch["image"])
        y = m(batch["image"].to(DEV))
    assert (y.cpu() - yo).abs().max().item() < 1e-3 * max(1.0, yo.abs().max().item())


def test_unetpp_train_step_matches_oracle():
    ora, m = _build("resnet18", 5)
    ora.train(); m.train()
    batch = synthetic_batch(2, 3, 128, 5, 5)
    yo = ora(batch["image"])
    lo = dice_loss_multiclass(yo, batch["mask"].squeeze(1).long())
    lo.backward()
    y = m(batch["image"].to(DEV))
    loss = gnn.DiceLoss()(y, batch["mask"].to(DEV))
    loss.backward()
    assert (y.detach().cpu() - yo.detach()).abs().max().item() < 1e-3 * max(1.0, yo.abs().max().item())
    assert abs(loss.item() - lo.item()) < 1e-5
    ref = dict(ora.named_parameters())
    bad = []
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        err, rn = (p.grad.cpu() - ref[n].grad).norm().item(), ref[n].grad.norm().item()
        if err > 3e-2 * rn + 2e-6:
            bad.append((n, err, rn))
    assert not bad, bad[:8]
    rb = dict(ora.named_buffers())
    for n, b in m.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            assert torch.allclose(b.cpu(), rb[n], atol=1e-4, rtol=1e-4), n
        if n.endswith("num_batches_tracked"):      # advanced by ONE multi-tensor launch at the end of the forward (gnn.counter_batch)
            assert int(b) == int(rb[n]) == 1, n


def test_unetpp_train_bf16_descends():
    ora, m = _build("resnet18", 9)
    m.train()
    batch = synthetic_batch(2, 3, 128, 5, 9)
    x, y = batch["image"].to(DEV), batch["mask"].to(DEV)
    opt = gnn.FusedAdam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = gnn.DiceLoss()(m(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


def test_unetpp_512_bf16_eval_and_train_step_match_oracle():
    """configs[0] at the BENCHMARKED shape and dtype: UNet++ / ResNet18 on 512 x 512 RGB under bf16 autocast -- eval logits
    and one training step (loss, every parameter gradient in the relative L2 norm, BatchNorm running statistics) against
    the f32 oracle on the CPU.  The 512^2 / 256^2 decoder stages run the direct narrow 3x3 kernel and the tap-packed
    256 x 64 tiles that smaller test images never reach."""
    ora, m = _build("resnet18", 21)
    batch = synthetic_batch(2, 3, 512, 5, 21)
    x, yt = batch["image"].to(DEV), batch["mask"].to(DEV)
    ora.eval(); m.eval()
    with torch.no_grad():
        yo = ora(batch["image"])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = m(x)
    sc = yo.abs().max().item()
    assert (yb.float().cpu() - yo).abs().max().item() < 0.06 * sc
    top2 = yo.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 0.06 * sc
    assert (gnn.predict_mask(yb.float()).cpu() == yo.argmax(1))[decided].all()
    ora.train(); m.train()
    lo = dice_loss_multiclass(ora(batch["image"]), batch["mask"].squeeze(1).long())
    lo.backward()
    ref = {n: p.grad.clone() for n, p in ora.named_parameters()}
    rb = {n: b.clone() for n, b in ora.named_buffers()}
    # second reference: the SAME oracle under torch's own bf16 autocast on the CPU.  This network at a random initialisation
    # is ill-conditioned in bf16 whoever computes it: the Dice gradient is nearly constant over the pixels of a class, every
    # train-mode BatchNorm backward subtracts that common part (dy - mean(dy) - xhat mean(dy xhat)), and the bf16 rounding of
    # dy before the subtraction is of the size of what remains -- torch's CPU bf16 step is 52 % (median, relative L2) away
    # from its own f32 step, growing from 2 % at the head to 50 % at the encoder.  The build is held to the f32 oracle where
    # that is meaningful (head, loss, statistics; the f32 build at this shape is within 1 % in test_unetpp_train_step...)
    # and to "not further from f32 than torch's bf16" everywhere else.
    ora.zero_grad(set_to_none=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y16 = ora(batch["image"])
    dice_loss_multiclass(y16.float(), batch["mask"].squeeze(1).long()).backward()
    ref16 = {n: p.grad.float() for n, p in ora.named_parameters()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = gnn.DiceLoss()(m(x), yt)
    loss.backward()
    assert abs(loss.item() - lo.item()) < 2e-2 * abs(lo.item())
    rels, rels16 = [], []
    for n, p in m.named_parameters():
        g, r = p.grad.float().cpu(), ref[n]
        assert torch.isfinite(g).all(), n
        if r.norm().item() > 1e-6:
            rels.append((float((g - r).norm() / r.norm()), n))
            rels16.append((float((ref16[n] - r).norm() / r.norm()), n))
    by_name, by_name16 = {n: e for e, n in rels}, {n: e for e, n in rels16}
    rels.sort(); rels16.sort()
    med, p90, worst = rels[len(rels) // 2][0], rels[int(0.9 * len(rels))][0], rels[-1]
    med16, p9016 = rels16[len(rels16) // 2][0], rels16[int(0.9 * len(rels16))][0]
    print(f"UNet++ 512^2 bf16 train step: gradient relative L2 vs the f32 oracle: median {med:.3f}, 90 % {p90:.3f}, worst {worst}; "
          f"torch CPU bf16 autocast vs the same: median {med16:.3f}, 90 % {p9016:.3f}; head {by_name['segmentation_head.0.weight']:.4f} "
          f"(torch bf16 {by_name16['segmentation_head.0.weight']:.4f})")
    assert med <= 1.15 * med16 + 0.01 and p90 <= 1.15 * p9016 + 0.01, (med, med16, p90, p9016)
    for n in ("segmentation_head.0.weight", "segmentation_head.0.bias", "decoder.blocks.x_0_4.conv2.1.weight"):
        assert by_name[n] <= max(0.08, 1.5 * by_name16[n]), (n, by_name[n], by_name16[n])
    for n, b in m.named_buffers():
        if n.endswith("running_mean"):
            assert torch.allclose(b.cpu(), rb[n], atol=2e-2, rtol=5e-2), n


@pytest.mark.parametrize("enc,size", [("resnet50", 64), ("resnext50_32x4d", 64), ("resnext101_32x8d", 64)])
def test_unetpp_bottleneck_and_resnext_encoders_eval(enc, size):
    """Bottleneck ResNets and ResNeXts (grouped 3x3 run as block-diagonal dense filters, gdlhip.cnn.mark_groups) -- incl. the
    encoder of the reference's shipped config, resnext101_32x8d: eval logits in f32 (1e-3) and bf16 vs the oracle."""
    ora, m = _build(enc, 17)
    ora.eval(); m.eval()
    batch = synthetic_batch(2, 3, size, 5, 17)
    x = batch["image"].to(DEV)
    with torch.no_grad():
        yo = ora(batch["image"])
        y = m(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yb = m(x)
    sc = max(1.0, yo.abs().max().item())
    assert y.shape == yo.shape == (2, 5, size, size)
    assert (y.cpu() - yo).abs().max().item() < 1e-3 * sc
    assert (yb.float().cpu() - yo).abs().max().item() < 0.08 * yo.abs().max().item()


@pytest.mark.parametrize("enc", ["resnet50", "resnext50_32x4d"])
def test_unetpp_bottleneck_train_step_matches_oracle(enc):
    """One f32 training step through the Bottleneck / grouped-3x3 encoder: loss, every parameter gradient (the grouped
    filters' gradients are the block diagonals of the dense ones) and the BatchNorm buffers vs the oracle.  Fifty layers of
    train-mode BatchNorm on two images are ill-conditioned in f32 for anyone -- the oracle's own f32 gradients sit 1.9 %
    (median, relative L2) from its f64 gradients (ResNet18: 0.4 %) -- so the truth is the oracle in f64 and the build is held to
    "no further from it than 1.5 x the f32 oracle"."""
    ora, m = _build(enc, 23)
    ora.train(); m.train()
    batch = synthetic_batch(2, 3, 128, 5, 23)
    tgt = batch["mask"].squeeze(1).long()
    yo = ora(batch["image"])
    lo = dice_loss_multiclass(yo, tgt)
    lo.backward()
    g32 = {n: p.grad.double() for n, p in ora.named_parameters()}
    rb = {n: b.clone() for n, b in ora.named_buffers()}
    o64 = OracleUnetPlusPlus(enc, 3, 5)
    o64.load_state_dict(procedural_state_dict(o64, 23))
    o64 = o64.double().train()
    dice_loss_multiclass(o64(batch["image"].double()), tgt).backward()
    g64 = {n: p.grad for n, p in o64.named_parameters()}
    y = m(batch["image"].to(DEV))
    loss = gnn.DiceLoss()(y, batch["mask"].to(DEV))
    loss.backward()
    assert (y.detach().cpu() - yo.detach()).abs().max().item() < 1e-3 * max(1.0, yo.abs().max().item())
    assert abs(loss.item() - lo.item()) < 1e-5
    e_build, e_o32 = [], []
    for n, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == g64[n].shape, n
        e_build.append(((p.grad.double().cpu() - g64[n]).norm() / (g64[n].norm() + 1e-30)).item())
        e_o32.append(((g32[n] - g64[n]).norm() / (g64[n].norm() + 1e-30)).item())
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    print(f"{enc}: gradient error vs the f64 oracle, relative L2: build median {med(e_build):.4f} max {max(e_build):.4f}; "
          f"f32 oracle median {med(e_o32):.4f} max {max(e_o32):.4f}")
    assert med(e_build) <= 1.5 * med(e_o32) + 1e-3 and max(e_build) <= 2.0 * max(e_o32) + 1e-3
    for n, b in m.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            assert torch.allclose(b.cpu(), rb[n], atol=1e-4, rtol=1e-4), n
