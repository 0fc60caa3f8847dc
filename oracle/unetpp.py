"""Oracle restatement of ``smp.UnetPlusPlus`` with a torchvision ResNet BasicBlock encoder -- the model the
reference instantiates at tasks_with_models/segmentation_unetplus.py:126-131.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

**Parity unpinned.**  Neither segmentation-models-pytorch 0.5.0 (uv.lock.cpu:3125-3126) nor torchvision 0.23.0
(uv.lock.cpu:3464-3465) is under /root/reference or installed in this image, and the reference's own test for this
path (tests/test_notebooks_00quickstart.py:98-118) asserts nothing numeric.  This file restates their published
algorithms:

* torchvision ``models/resnet.py``: ``BasicBlock`` (conv3x3-BN-ReLU, conv3x3-BN, + identity / 1x1-stride-s
  downsample, ReLU), ``Bottleneck`` (1x1, grouped 3x3 carrying the stride, 1x1 to 4 x planes; resnet50 / 101,
  resnext50_32x4d, resnext101_32x8d -- the encoder the reference's shipped config names), ``ResNet._make_layer``, stem
  conv 7x7/2 + BN + ReLU + max-pool 3x3/2 (pad 1);
* smp ``encoders/resnet.py`` ``ResNetEncoder.forward`` feature list (identity, stem, layer1..4; ``fc`` removed);
* smp ``decoders/unetplusplus/decoder.py``: ``DecoderBlock`` (nearest x2, concat skip, 2 x Conv2dReLU) and the
  dense grid ``x_{depth}_{layer}``; ``base/modules.py`` ``Conv2dReLU`` = Conv2d(bias=False) + BatchNorm2d + ReLU;
* smp ``base/heads.py`` ``SegmentationHead`` = Conv2d(16, classes, 3, padding=1) (+ identity upsampling/activation).

The only external anchor is the parameter count the reference's notebook prints for resnet34 / 2 classes
(26.1 M, notebooks/00_quickstart.ipynb:572), checked in tests/test_oracle_unetpp.py.  State-dict keys follow smp
(``encoder.layer1.0.conv1.weight``, ``decoder.blocks.x_0_0.conv1.0.weight``, ``segmentation_head.0.weight``).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor, nn

RESNET_LAYERS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}
# torchvision models/resnet.py constructors with Bottleneck blocks: (layers, groups, width_per_group).  The reference's SHIPPED
# UNet++ config names resnext101_32x8d (configs/unetplus_config_RGB.yaml:37).
BOTTLENECK_SPECS = {"resnet50": ([3, 4, 6, 3], 1, 64), "resnet101": ([3, 4, 23, 3], 1, 64),
                    "resnext50_32x4d": ([3, 4, 6, 3], 32, 4), "resnext101_32x8d": ([3, 4, 23, 3], 32, 8)}


class BasicBlock(nn.Module):
    def __init__(self, inplanes: int, planes: int, stride: int = 1) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x: Tensor) -> Tensor:
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + identity)


class Bottleneck(nn.Module):
    """torchvision models/resnet.py Bottleneck (v1.5: the stride sits on the 3x3 convolution), expansion 4;
    ``width = int(planes * base_width / 64) * groups`` and the 3x3 convolution is grouped (ResNeXt)."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, groups: int = 1, base_width: int = 64) -> None:
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x: Tensor) -> Tensor:
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + identity)


class ResNetEncoder(nn.Module):
    def __init__(self, name: str = "resnet18", in_channels: int = 3) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        if name in BOTTLENECK_SPECS:
            layers, groups, base_width = BOTTLENECK_SPECS[name]
            for i, (planes, blocks) in enumerate(zip([64, 128, 256, 512], layers)):
                layer = []
                for j in range(blocks):
                    layer.append(Bottleneck(inplanes, planes, (1 if i == 0 else 2) if j == 0 else 1, groups, base_width))
                    inplanes = planes * 4
                setattr(self, f"layer{i + 1}", nn.Sequential(*layer))
            self.out_channels = (in_channels, 64, 256, 512, 1024, 2048)
            return
        for i, (planes, blocks) in enumerate(zip([64, 128, 256, 512], RESNET_LAYERS[name])):
            layer = []
            for j in range(blocks):
                layer.append(BasicBlock(inplanes, planes, (1 if i == 0 else 2) if j == 0 else 1))
                inplanes = planes
            setattr(self, f"layer{i + 1}", nn.Sequential(*layer))
        self.out_channels = (in_channels, 64, 64, 128, 256, 512)

    def forward(self, x: Tensor) -> list[Tensor]:
        feats = [x]
        x = F.relu(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.layer1(F.max_pool2d(x, 3, 2, 1))
        feats.append(x)
        for i in (2, 3, 4):
            x = getattr(self, f"layer{i}")(x)
            feats.append(x)
        return feats


def _conv2d_relu(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DecoderBlock(nn.Module):
    def __init__(self, in_channels: int, skip_channels: int, out_channels: int) -> None:
        super().__init__()
        self.conv1 = _conv2d_relu(in_channels + skip_channels, out_channels)
        self.conv2 = _conv2d_relu(out_channels, out_channels)

    def forward(self, x: Tensor, skip: Tensor | None = None) -> Tensor:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class UnetPlusPlusDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=(256, 128, 64, 32, 16)) -> None:
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        self.in_channels = [enc[0], *decoder_channels[:-1]]
        self.skip_channels = [*enc[1:], 0]
        self.out_channels = list(decoder_channels)
        blocks = {}
        for layer in range(len(self.in_channels) - 1):
            for depth in range(layer + 1):
                if depth == 0:
                    cin, skip, cout = self.in_channels[layer], self.skip_channels[layer] * (layer + 1), self.out_channels[layer]
                else:
                    cout = self.skip_channels[layer]
                    skip = self.skip_channels[layer] * (layer + 1 - depth)
                    cin = self.skip_channels[layer - 1]
                blocks[f"x_{depth}_{layer}"] = DecoderBlock(cin, skip, cout)
        blocks[f"x_0_{len(self.in_channels) - 1}"] = DecoderBlock(self.in_channels[-1], 0, self.out_channels[-1])
        self.blocks = nn.ModuleDict(blocks)
        self.depth = len(self.in_channels) - 1

    def forward(self, feats: list[Tensor]) -> Tensor:
        feats = feats[1:][::-1]
        dense = {}
        for layer in range(len(self.in_channels) - 1):
            for depth in range(self.depth - layer):
                if layer == 0:
                    dense[f"x_{depth}_{depth}"] = self.blocks[f"x_{depth}_{depth}"](feats[depth], feats[depth + 1])
                else:
                    li = depth + layer
                    cat = [dense[f"x_{idx}_{li}"] for idx in range(depth + 1, li + 1)]
                    cat = torch.cat([*cat, feats[li + 1]], dim=1)
                    dense[f"x_{depth}_{li}"] = self.blocks[f"x_{depth}_{li}"](dense[f"x_{depth}_{li - 1}"], cat)
        dense[f"x_0_{self.depth}"] = self.blocks[f"x_0_{self.depth}"](dense[f"x_0_{self.depth - 1}"])
        return dense[f"x_0_{self.depth}"]


class UnetPlusPlus(nn.Module):
    def __init__(self, encoder_name: str = "resnet34", in_channels: int = 3, classes: int = 1) -> None:
        super().__init__()
        self.encoder = ResNetEncoder(encoder_name, in_channels)
        self.decoder = UnetPlusPlusDecoder(self.encoder.out_channels)
        self.segmentation_head = nn.Sequential(nn.Conv2d(16, classes, 3, padding=1), nn.Identity(), nn.Identity())

    def forward(self, x: Tensor) -> Tensor:
        return self.segmentation_head(self.decoder(self.encoder(x)))
