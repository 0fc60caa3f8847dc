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
