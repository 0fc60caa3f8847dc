"""UNet++ with a ResNet (BasicBlock or Bottleneck / ResNeXt) encoder on MI355X: drop-in for the ``smp.UnetPlusPlus`` instance the reference
builds at tasks_with_models/segmentation_unetplus.py:126-131 (same constructor keywords, same state-dict keys as
segmentation-models-pytorch 0.5.0 / torchvision, so its checkpoints load).

Everything runs NHWC in the compute dtype on the implicit-GEMM MFMA kernel:
* stem 7x7/2 on the raw bands = space-to-depth re-layout (4 x 4 pixel blocks) + four sub-pixel-phase 3x3 implicit-GEMM
  convolutions + BN + ReLU (no im2col matrix), then the 3x3/2 max-pool kernel;
* BasicBlock = conv3x3-BN-ReLU, conv3x3-BN, (+ 1x1/s downsample-BN), residual add + ReLU (one kernel in
  training; folded into the second conv's epilogue in eval); Bottleneck = 1x1, (grouped) 3x3 with the stride, 1x1 to
  4 x planes, same tail -- resnet50 / 101, resnext50_32x4d and resnext101_32x8d, the encoder of the reference's shipped
  config (configs/unetplus_config_RGB.yaml:37);
* DecoderBlock = nearest x2 of the input written straight into the dense-skip concat buffer, then two
  conv3x3-BN-ReLU; the 32- and 16-channel stages run at their true width (channel-tail kernels, gdlhip.cnn);
* head = 3x3 conv to ``classes`` -> NCHW f32 logits.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from geo_deep_learning.models.utils import _cl_conv
from gdlhip import cnn, ops
from gdlhip import nn as gnn

RESNET_LAYERS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}
# torchvision's Bottleneck ResNets / ResNeXts: (layers, groups, width_per_group).  resnext101_32x8d is the encoder of the
# reference's shipped config (configs/unetplus_config_RGB.yaml:37).
BOTTLENECK_SPECS = {"resnet50": ([3, 4, 6, 3], 1, 64), "resnet101": ([3, 4, 23, 3], 1, 64),
                    "resnext50_32x4d": ([3, 4, 6, 3], 32, 4), "resnext101_32x8d": ([3, 4, 23, 3], 32, 8)}


def _cl_conv_g(cin: int, cout: int, k: int, *, padding: int, groups: int) -> nn.Conv2d:
    """Grouped Conv2d(bias=False) parameter container, stored channels-last like _cl_conv ([N, R, S, C / groups] in memory)."""
    conv = nn.Conv2d(cin, cout, k, padding=padding, groups=groups, bias=False)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


class BasicBlock(nn.Module):
    """torchvision models/resnet.py BasicBlock (parameter names conv1/bn1/conv2/bn2/downsample.{0,1})."""

    def __init__(self, inplanes: int, planes: int, stride: int = 1) -> None:
        super().__init__()
        self.stride = stride
        self.conv1 = _cl_conv(inplanes, planes, 3, padding=1, bias=False)
        self.conv1.stride = (stride, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _cl_conv(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            ds = _cl_conv(inplanes, planes, 1, padding=0, bias=False)
            ds.stride = (stride, stride)
            self.downsample = nn.Sequential(ds, nn.BatchNorm2d(planes))

    def forward_nhwc(self, x: Tensor) -> Tensor:
        identity = x
        if self.downsample is not None:
            identity = cnn.conv_bn(x, self.downsample[0].weight, self.downsample[1], stride=self.stride, relu=False)
        out = cnn.conv_bn(x, self.conv1.weight, self.bn1, stride=self.stride, pad=1)
        return cnn.conv_bn(out, self.conv2.weight, self.bn2, pad=1, resid=identity)


class Bottleneck(nn.Module):
    """torchvision models/resnet.py Bottleneck (v1.5: stride on the 3x3), parameter names conv1..3 / bn1..3 /
    downsample.{0,1}; ``groups`` > 1 = ResNeXt's grouped 3x3 (gdlhip.cnn.mark_groups: run as a block-diagonal dense filter)."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, groups: int = 1, base_width: int = 64) -> None:
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.stride, self.groups = stride, groups
        self.conv1 = _cl_conv(inplanes, width, 1, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _cl_conv_g(width, width, 3, padding=1, groups=groups)
        self.conv2.stride = (stride, stride)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = _cl_conv(width, planes * 4, 1, padding=0, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or inplanes != planes * 4:
            ds = _cl_conv(inplanes, planes * 4, 1, padding=0, bias=False)
            ds.stride = (stride, stride)
            self.downsample = nn.Sequential(ds, nn.BatchNorm2d(planes * 4))

    def forward_nhwc(self, x: Tensor) -> Tensor:
        identity = x
        if self.downsample is not None:
            identity = cnn.conv_bn(x, self.downsample[0].weight, self.downsample[1], stride=self.stride, relu=False)
        cnn.mark_groups(self.conv2.weight, self.groups)      # (the mark lives on the Parameter object)
        out = cnn.conv_bn(x, self.conv1.weight, self.bn1)
        out = cnn.conv_bn(out, self.conv2.weight, self.bn2, stride=self.stride, pad=1)
        return cnn.conv_bn(out, self.conv3.weight, self.bn3, resid=identity)


class ResNetEncoder(nn.Module):
    """smp encoders/resnet.py ResNetEncoder (depth 5): features at strides 1, 2, 4, 8, 16, 32."""

    def __init__(self, name: str = "resnet18", in_channels: int = 3) -> None:
        super().__init__()
        if name not in RESNET_LAYERS and name not in BOTTLENECK_SPECS:
            msg = (f"gdlhip UnetPlusPlus: encoder {name!r} is not built (BasicBlock ResNets {sorted(RESNET_LAYERS)}, "
                   f"Bottleneck ResNets / ResNeXts {sorted(BOTTLENECK_SPECS)})")
            raise NotImplementedError(msg)
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)     # standard OIHW layout: consumed flat
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

    def forward_nhwc(self, img: Tensor) -> list[Tensor]:
        """NCHW f32 image -> the five NHWC feature maps (strides 2..32) in the compute dtype."""
        cd = gnn.compute_dtype()
        # im2col-free stem: the image is re-laid into 4 x 4 pixel blocks (16 C channels) and the 7x7 / 2 convolution runs as
        # four sub-pixel-phase 3x3 convolutions on that map (gdlhip.cnn.mark_stem)
        if cnn.STEM_IM2COL:      # A/B: strided patchify (an im2col matrix of 49 / 4 x the image) + GEMM
            b, c, h, w = img.shape
            gh, gw = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
            kpad = cnn.pad_to(c * 49, cnn.chunk(cd))
            cols = ops.patchify(ops.image_f32(img, "UnetPlusPlus"), 7, 3, gh, gw, kpad, cd, stride=2).view(b, gh, gw, kpad)
            cnn.mark_flat(self.conv1.weight)
            x = cnn.conv_bn(cols, self.conv1.weight, self.bn1)
        else:
            xs = cnn.space_to_depth_image(ops.image_f32(img, "UnetPlusPlus"), cd)
            cnn.mark_stem(self.conv1.weight, 2, 3)
            x = cnn.conv_bn(xs, self.conv1.weight, self.bn1)
        feats = [x]
        x = cnn.maxpool3x3s2(x)
        for i in (1, 2, 3, 4):
            for blk in getattr(self, f"layer{i}"):
                x = blk.forward_nhwc(x)
            feats.append(x)
        return feats

    def forward(self, img: Tensor) -> list[Tensor]:
        return [img, *[ops.as_nchw(f) for f in self.forward_nhwc(img)]]


class Conv2dReLU(nn.Sequential):
    """smp base/modules.py Conv2dReLU: Conv2d(bias=False) + BatchNorm2d + ReLU (keys ``0.weight``, ``1.*``)."""

    def __init__(self, cin: int, cout: int) -> None:
        super().__init__(_cl_conv(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward_nhwc(self, x: Tensor) -> Tensor:
        return cnn.conv_bn(x, self[0].weight, self[1], pad=1)


class DecoderBlock(nn.Module):
    """smp decoders/unetplusplus/decoder.py DecoderBlock (attention type None)."""

    def __init__(self, in_channels: int, skip_channels: int, out_channels: int) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.conv1 = Conv2dReLU(in_channels + skip_channels, out_channels)
        self.conv2 = Conv2dReLU(out_channels, out_channels)

    def forward_nhwc(self, x: Tensor, skips: list[Tensor]) -> Tensor:
        if x.shape[-1] != self.in_channels and skips:
            msg = "a channel-padded tensor cannot be concatenated with skips"   # only x_0_4 (no skips) sees padding
            raise ValueError(msg)
        return self.conv2.forward_nhwc(self.conv1.forward_nhwc(cnn.up_cat(x, skips)))


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
                    cin, skip, cout = (self.in_channels[layer], self.skip_channels[layer] * (layer + 1),
                                       self.out_channels[layer])
                else:
                    cout = self.skip_channels[layer]
                    skip = self.skip_channels[layer] * (layer + 1 - depth)
                    cin = self.skip_channels[layer - 1]
                blocks[f"x_{depth}_{layer}"] = DecoderBlock(cin, skip, cout)
        blocks[f"x_0_{len(self.in_channels) - 1}"] = DecoderBlock(self.in_channels[-1], 0, self.out_channels[-1])
        self.blocks = nn.ModuleDict(blocks)
        self.depth = len(self.in_channels) - 1

    def forward_nhwc(self, feats: list[Tensor]) -> Tensor:
        """feats: the encoder's five NHWC maps, shallow to deep."""
        feats = feats[::-1]
        dense: dict[str, Tensor] = {}
        for layer in range(len(self.in_channels) - 1):
            for depth in range(self.depth - layer):
                if layer == 0:
                    dense[f"x_{depth}_{depth}"] = self.blocks[f"x_{depth}_{depth}"].forward_nhwc(
                        feats[depth], [feats[depth + 1]])
                else:
                    li = depth + layer
                    skips = [dense[f"x_{idx}_{li}"] for idx in range(depth + 1, li + 1)] + [feats[li + 1]]
                    dense[f"x_{depth}_{li}"] = self.blocks[f"x_{depth}_{li}"].forward_nhwc(
                        dense[f"x_{depth}_{li - 1}"], skips)
        return self.blocks[f"x_0_{self.depth}"].forward_nhwc(dense[f"x_0_{self.depth - 1}"], [])


class UnetPlusPlus(nn.Module):
    """``smp.UnetPlusPlus(encoder_name, in_channels, encoder_weights, classes)`` -> NCHW f32 logits."""

    def __init__(self, encoder_name: str = "resnet34", encoder_depth: int = 5, encoder_weights: str | None = "imagenet",
                 decoder_channels=(256, 128, 64, 32, 16), in_channels: int = 3, classes: int = 1,
                 activation=None, aux_params=None, **kwargs: object) -> None:
        super().__init__()
        if encoder_depth != 5 or activation is not None or aux_params is not None or kwargs:
            msg = "gdlhip UnetPlusPlus implements the reference's configuration (depth 5, no activation / aux head)"
            raise NotImplementedError(msg)
        self.encoder = ResNetEncoder(encoder_name, in_channels)
        if encoder_weights is not None:
            self._load_cached_encoder_weights(encoder_name, encoder_weights, in_channels)
        self.decoder = UnetPlusPlusDecoder(self.encoder.out_channels, tuple(decoder_channels))
        self.segmentation_head = nn.Sequential(_cl_conv(decoder_channels[-1], classes, 3, padding=1, bias=True),
                                               nn.Identity(), nn.Identity())

    def _load_cached_encoder_weights(self, name: str, weights: str, in_channels: int) -> None:
        """``encoder_weights="imagenet"`` (configs/unetplus_config_RGB.yaml:39): smp downloads torchvision's checkpoint into the
        torch-hub cache; this build never downloads -- it reads ``<hub>/checkpoints/<name>-*.pth`` when a previous (networked)
        run or the operator put it there, and says where it looked otherwise."""
        import glob
        import logging
        import os
        pattern = os.path.join(torch.hub.get_dir(), "checkpoints", f"{name}-*.pth")
        found = sorted(glob.glob(pattern))
        if os.path.isfile(str(weights)):                   # an explicit checkpoint path instead of a weight-set name
            found, weights = [str(weights)], "imagenet"
        elif len(found) > 1:
            # several cached variants (torchvision's IMAGENET1K_V1 / V2, the pre-0.13 files): "the last one in lexical order" is
            # not a choice.  smp's `imagenet` weights are torchvision's V1 set: take that file, or ask for an explicit path
            v1 = {"resnet18": ("f37072fd", "5c106cde"), "resnet34": ("b627a593", "333f7ec4"), "resnet50": ("0676ba61", "19c8e357"),
                  "resnet101": ("63fe2227", "5d3b4d8f"), "resnext50_32x4d": ("7cdf4587",), "resnext101_32x8d": ("8ba56ff5",)}
            pick = [f for h in v1.get(name, ()) for f in found if f.endswith(f"{name}-{h}.pth")]
            if not pick:
                msg = (f"encoder_weights={weights!r}: {len(found)} cached checkpoints match {pattern} and none is torchvision's "
                       f"IMAGENET1K_V1 file; pass the file to use as encoder_weights=<path>: {found}")
                raise RuntimeError(msg)
            found = pick[:1]
        if weights != "imagenet" or not found or in_channels != 3:
            msg = (f"encoder_weights={weights!r}: the reference lets smp download torchvision's {name} checkpoint; this build has no "
                   f"network and found {'no file' if not found else 'a file but in_channels != 3'} at {pattern}. Pass encoder_weights=None "
                   "and load a checkpoint with load_state_dict, or place the torchvision checkpoint there")
            raise RuntimeError(msg)
        sd = torch.load(found[-1], map_location="cpu", weights_only=True)
        meta = getattr(sd, "_metadata", None)
        sd = type(sd)((k, v) for k, v in sd.items() if not k.startswith("fc."))
        if meta is not None:
            sd._metadata = meta                            # the version records load_state_dict's compatibility hooks read
        self.encoder.load_state_dict(sd, strict=True)
        logging.getLogger(__name__).info("UNet++ encoder %s: loaded %s", name, found[-1])

    def forward(self, x: Tensor) -> Tensor:
        if x.shape[2] % 32 or x.shape[3] % 32:
            msg = f"Wrong input shape height={x.shape[2]}, width={x.shape[3]}: must be divisible by 32"   # smp's check
            raise RuntimeError(msg)
        with gnn.counter_batch():       # the BatchNorm step counters of the pass advance in one launch
            dec = self.decoder.forward_nhwc(self.encoder.forward_nhwc(x))
            return cnn.logits_nchw(cnn.conv_bias(dec, self.segmentation_head[0]))
