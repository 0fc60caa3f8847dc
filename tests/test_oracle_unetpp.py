"""UNet++ oracle self-consistency (parity unpinned -- smp / torchvision are not in the image).
The one external anchor is the parameter count the reference's notebook prints (notebooks/00_quickstart.ipynb:572:
26.1 M for resnet34, 2 classes); the rest checks structure: smp state-dict key names, channel plan, output shape."""

import torch

from oracle.unetpp import UnetPlusPlus


def test_param_count_matches_the_reference_notebook():
    m = UnetPlusPlus("resnet34", 3, 2)
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 26.1e6) < 0.05e6, n


def test_structure_and_shapes():
    m = UnetPlusPlus("resnet18", 3, 5).eval()
    keys = set(m.state_dict())
    for k in ("encoder.conv1.weight", "encoder.layer2.0.downsample.0.weight", "encoder.layer4.1.bn2.running_var",
              "decoder.blocks.x_0_0.conv1.0.weight", "decoder.blocks.x_3_3.conv2.1.bias", "decoder.blocks.x_0_4.conv1.0.weight",
              "segmentation_head.0.bias"):
        assert k in keys, k
    b = m.decoder.blocks
    assert b["x_0_0"].conv1[0].weight.shape == (256, 768, 3, 3)
    assert b["x_0_3"].conv1[0].weight.shape == (32, 320, 3, 3)
    assert b["x_1_3"].conv1[0].weight.shape == (64, 256, 3, 3)
    assert b["x_0_4"].conv1[0].weight.shape == (16, 32, 3, 3)
    assert len(b) == 11
    with torch.no_grad():
        y = m(torch.zeros(1, 3, 64, 64))
    assert y.shape == (1, 5, 64, 64)


def _hf_resnet(name):
    """An independent implementation of the same encoder: ``transformers`` ResNet with BasicBlock ("basic") layers --
    the torchvision resnet18 / resnet34 topology (stem 7x7/2 + BN + ReLU + max-pool 3x3/2, stages of two 3x3 convs with
    a 1x1 stride-2 projection shortcut at the start of stages 2-4, ReLU after the residual sum)."""
    import pytest
    transformers = pytest.importorskip("transformers")
    from oracle.unetpp import RESNET_LAYERS
    cfg = transformers.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512],
                                    depths=RESNET_LAYERS[name], layer_type="basic", hidden_act="relu",
                                    downsample_in_first_stage=False)
    return transformers.ResNetModel(cfg).eval()


def _copy_hf_to_oracle(hf, enc):
    """HF names -> torchvision / smp names (the oracle's)."""
    src = hf.state_dict()
    dst = {}

    def bn(dprefix, sprefix):
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            dst[f"{dprefix}.{k}"] = src[f"{sprefix}.normalization.{k}"]
    dst["conv1.weight"] = src["embedder.embedder.convolution.weight"]
    bn("bn1", "embedder.embedder")
    for si in range(4):
        j = 0
        while f"encoder.stages.{si}.layers.{j}.layer.0.convolution.weight" in src:
            s, d = f"encoder.stages.{si}.layers.{j}", f"layer{si + 1}.{j}"
            dst[f"{d}.conv1.weight"] = src[f"{s}.layer.0.convolution.weight"]
            bn(f"{d}.bn1", f"{s}.layer.0")
            dst[f"{d}.conv2.weight"] = src[f"{s}.layer.1.convolution.weight"]
            bn(f"{d}.bn2", f"{s}.layer.1")
            if f"{s}.shortcut.convolution.weight" in src:
                dst[f"{d}.downsample.0.weight"] = src[f"{s}.shortcut.convolution.weight"]
                bn(f"{d}.downsample.1", f"{s}.shortcut")
            j += 1
    missing, unexpected = enc.load_state_dict(dst, strict=True), None
    return missing, unexpected


def test_resnet_encoder_matches_an_independent_implementation():
    """Numeric cross-check of the encoder half of the UNet++ oracle (the way Dinov2 pins the ViT block): same weights
    in ``transformers``' ResNet and in oracle.unetpp.ResNetEncoder -> the same five feature maps (strides 2..32)."""
    import pytest
    from oracle.unetpp import ResNetEncoder
    for name in ("resnet18", "resnet34"):
        hf = _hf_resnet(name)
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for n, p in hf.named_parameters():                      # non-trivial BN affine / running statistics
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() == 4 else 0.5) + (1.0 if n.endswith("normalization.weight") else 0.0))
            for n, b in hf.named_buffers():
                if n.endswith("running_mean"):
                    b.copy_(torch.randn(b.shape, generator=g) * 0.1)
                elif n.endswith("running_var"):
                    b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        enc = ResNetEncoder(name, 3).eval()
        _copy_hf_to_oracle(hf, enc)
        assert len(enc.state_dict()) == len(hf.state_dict())           # every tensor has a counterpart
        x = torch.randn(2, 3, 96, 96, generator=g)
        with torch.no_grad():
            feats = enc(x)
            out = hf(x, output_hidden_states=True)
            stem = hf.embedder.embedder(x)                              # conv + BN + ReLU, before the max-pool
        assert torch.equal(feats[0], x)
        assert torch.allclose(feats[1], stem, atol=1e-5, rtol=1e-5)
        # hidden_states[0] = after the max-pool; [1..4] = the four stages
        assert [tuple(f.shape[1:]) for f in feats[1:]] == [(64, 48, 48), (64, 24, 24), (128, 12, 12), (256, 6, 6), (512, 3, 3)]
        for i in range(4):
            assert torch.allclose(feats[2 + i], out.hidden_states[1 + i], atol=2e-5, rtol=1e-4), (name, i)
    pytest.importorskip("transformers")


def test_smp_state_dict_key_list_is_exact():
    """The COMPLETE key list of smp 0.5.0 ``UnetPlusPlus("resnet34", classes=2)`` as its published code generates it
    (torchvision ResNet names under ``encoder.``, ``decoder.blocks.x_{depth}_{layer}.conv{1,2}.{0,1}`` for the eleven
    nested blocks, ``segmentation_head.0``): 36 encoder conv / downsample weights + 36 BN layers x 5 tensors ... --
    derived here from the naming rules, not from the oracle, and compared with the oracle's keys."""
    m = UnetPlusPlus("resnet34", 3, 2)
    want = ["encoder.conv1.weight"] + [f"encoder.bn1.{k}" for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    bnk = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")
    for li, blocks in enumerate([3, 4, 6, 3], start=1):
        for j in range(blocks):
            p = f"encoder.layer{li}.{j}"
            want += [f"{p}.conv1.weight"] + [f"{p}.bn1.{k}" for k in bnk] + [f"{p}.conv2.weight"] + [f"{p}.bn2.{k}" for k in bnk]
            if j == 0 and li > 1:
                want += [f"{p}.downsample.0.weight"] + [f"{p}.downsample.1.{k}" for k in bnk]
    # smp decoder: blocks[f"x_{depth_idx}_{layer_idx}"] for layer_idx in 0..3, depth_idx in 0..layer_idx, then x_0_4
    names = [f"x_{d}_{l}" for l in range(4) for d in range(l + 1)] + ["x_0_4"]
    for nme in names:
        for c in ("conv1", "conv2"):
            want += [f"decoder.blocks.{nme}.{c}.0.weight"] + [f"decoder.blocks.{nme}.{c}.1.{k}" for k in bnk]
    want += ["segmentation_head.0.weight", "segmentation_head.0.bias"]
    assert sorted(m.state_dict().keys()) == sorted(want)
    # channel plan of the dense grid (smp decoder.py: in = skip_channels[l-1] for depth > 0; skip = skip_ch[l] * (l + 1 - depth))
    b = m.decoder.blocks
    plan = {"x_0_0": (512 + 256, 256), "x_0_1": (256 + 128 * 2, 128), "x_1_1": (256 + 128, 128), "x_0_2": (128 + 64 * 3, 64),
            "x_1_2": (128 + 64 * 2, 64), "x_2_2": (128 + 64, 64), "x_0_3": (64 + 64 * 4, 32), "x_1_3": (64 + 64 * 3, 64),
            "x_2_3": (64 + 64 * 2, 64), "x_3_3": (64 + 64, 64), "x_0_4": (32, 16)}
    for k, (cin, cout) in plan.items():
        assert b[k].conv1[0].weight.shape[:2] == (cout, cin), (k, tuple(b[k].conv1[0].weight.shape))
        assert b[k].conv2[0].weight.shape[:2] == (cout, cout), k


def test_dense_skip_decoder_matches_the_published_recurrence():
    """The UNet++ decoder topology pinned by an independent construction: the nested dense skip
    pathways of Zhou et al. 2018, X[i, j] = H([X[i, 0], ..., X[i, j-1], Up(X[i+1, j-1])]) with X[i, 0] = encoder feature of
    stride 2^(i+1), written here as a functional loop over (j, i) that never touches oracle/unetpp.py's decoder code -- only
    its weights, looked up under smp's key names.  smp names node X[i, j] `x_{4-i-j}_{3-i}`, concatenates the upsampled
    deeper node first and the same-level nodes in descending j, and appends one skip-less block `x_0_4` that brings
    X[0, 4] to full resolution.  A wrong topology or key mapping cannot pass: the channel counts of the 22 convolutions
    only line up for this wiring, and the outputs must agree to 1e-5."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    m = UnetPlusPlus("resnet18", 3, 5).eval()
    with torch.no_grad():
        for name, t in m.state_dict().items():      # random weights AND random BatchNorm statistics
            if name.endswith("running_var"):
                t.copy_(torch.rand_like(t) + 0.5)
            elif name.endswith(("running_mean", "bias")):
                t.copy_(torch.randn_like(t) * 0.1)
            elif name.endswith("weight") and t.dim() == 1:
                t.copy_(torch.rand_like(t) + 0.5)
    sd = m.state_dict()
    x = torch.randn(2, 3, 64, 96)
    with torch.no_grad():
        e = m.encoder(x)                             # [x, e1 (1/2), e2 (1/4), e3 (1/8), e4 (1/16), e5 (1/32)]

        def conv_bn_relu(t, prefix):
            t = F.conv2d(t, sd[f"{prefix}.0.weight"], padding=1)
            t = F.batch_norm(t, sd[f"{prefix}.1.running_mean"], sd[f"{prefix}.1.running_var"], sd[f"{prefix}.1.weight"],
                             sd[f"{prefix}.1.bias"], training=False, eps=1e-5)
            return F.relu(t)

        def H(t, node):                              # smp DecoderBlock after the concat: two Conv2dReLU
            return conv_bn_relu(conv_bn_relu(t, f"decoder.blocks.{node}.conv1"), f"decoder.blocks.{node}.conv2")

        def up(t):
            return F.interpolate(t, scale_factor=2.0, mode="nearest")

        X = {(i, 0): e[i + 1] for i in range(5)}
        for j in range(1, 5):
            for i in range(0, 5 - j):
                cat = torch.cat([up(X[i + 1, j - 1])] + [X[i, jj] for jj in range(j - 1, -1, -1)], dim=1)
                X[i, j] = H(cat, f"x_{4 - i - j}_{3 - i}")
        out = H(up(X[0, 4]), "x_0_4")
        ref = m.decoder(e)
        assert out.shape == ref.shape == (2, 16, 64, 96)
        assert (out - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1.0)
        # and through the head, against the whole model
        logits = F.conv2d(out, sd["segmentation_head.0.weight"], sd["segmentation_head.0.bias"], padding=1)
        assert torch.allclose(logits, m(x), atol=1e-5)


def test_bottleneck_encoders_have_torchvision_parameter_counts():
    """Bottleneck ResNets / ResNeXts (the reference's shipped config names resnext101_32x8d,
    configs/unetplus_config_RGB.yaml:37): the encoder's parameter count equals torchvision's published total for the
    classification model minus its 2048 x 1000 + 1000 ``fc`` -- 25,557,032 / 44,549,160 / 25,028,904 / 88,791,336 -- which
    fixes widths, group counts, block counts and where the projection shortcuts sit; key names follow torchvision."""
    from oracle.unetpp import ResNetEncoder
    fc = 2048 * 1000 + 1000
    for name, total in (("resnet50", 25557032), ("resnet101", 44549160), ("resnext50_32x4d", 25028904), ("resnext101_32x8d", 88791336)):
        enc = ResNetEncoder(name, 3)
        assert sum(p.numel() for p in enc.parameters()) == total - fc, name
        assert enc.out_channels == (3, 64, 256, 512, 1024, 2048)
    enc = ResNetEncoder("resnext101_32x8d", 3)
    sd = enc.state_dict()
    assert tuple(sd["layer1.0.conv2.weight"].shape) == (256, 8, 3, 3)          # 32 groups x 8 channels
    assert tuple(sd["layer4.2.conv2.weight"].shape) == (2048, 64, 3, 3)
    assert tuple(sd["layer3.22.conv3.weight"].shape) == (1024, 1024, 1, 1) and "layer3.23.conv1.weight" not in sd
    assert tuple(sd["layer2.0.downsample.0.weight"].shape) == (512, 256, 1, 1) and "layer2.1.downsample.0.weight" not in sd
    m = UnetPlusPlus("resnext50_32x4d", 3, 2).eval()
    with torch.no_grad():
        assert m(torch.zeros(1, 3, 64, 64)).shape == (1, 2, 64, 64)


def test_bottleneck_encoder_matches_an_independent_implementation():
    """resnet50 in ``transformers`` (layer_type="bottleneck", stride on the 3x3 like torchvision v1.5) and in the oracle, same
    weights -> the same feature maps; the grouped 3x3 of the ResNeXt variants is torch's own ``F.conv2d(groups=32)``."""
    import pytest
    transformers = pytest.importorskip("transformers")
    from oracle.unetpp import ResNetEncoder
    cfg = transformers.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                                    layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                                    downsample_in_bottleneck=False)
    hf = transformers.ResNetModel(cfg).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.03 if p.dim() == 4 else 0.3) + (1.0 if n.endswith("normalization.weight") else 0.0))
        for n, b in hf.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
    src, dst = hf.state_dict(), {}

    def bn(d, s_):
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            dst[f"{d}.{k}"] = src[f"{s_}.normalization.{k}"]
    dst["conv1.weight"] = src["embedder.embedder.convolution.weight"]
    bn("bn1", "embedder.embedder")
    for si, blocks in enumerate([3, 4, 6, 3]):
        for j in range(blocks):
            s_, d = f"encoder.stages.{si}.layers.{j}", f"layer{si + 1}.{j}"
            for c in range(3):
                dst[f"{d}.conv{c + 1}.weight"] = src[f"{s_}.layer.{c}.convolution.weight"]
                bn(f"{d}.bn{c + 1}", f"{s_}.layer.{c}")
            if f"{s_}.shortcut.convolution.weight" in src:
                dst[f"{d}.downsample.0.weight"] = src[f"{s_}.shortcut.convolution.weight"]
                bn(f"{d}.downsample.1", f"{s_}.shortcut")
    enc = ResNetEncoder("resnet50", 3).eval()
    enc.load_state_dict(dst, strict=True)
    assert len(enc.state_dict()) == len(src)
    x = torch.randn(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        feats = enc(x)
        out = hf(x, output_hidden_states=True)
    assert [tuple(f.shape[1:]) for f in feats[2:]] == [(256, 16, 16), (512, 8, 8), (1024, 4, 4), (2048, 2, 2)]
    for i in range(4):
        ref = out.hidden_states[1 + i]
        assert torch.allclose(feats[2 + i], ref, atol=2e-4 * ref.abs().max().item(), rtol=1e-3), i
