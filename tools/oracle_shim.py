"""sys.modules shim so the REAL reference imports in this container (golden generation only).

The reference's hot-path files need four timm symbols and (for a dead class) kornia,
neither of which is installed here.  The shim's ``Block`` is
the oracle's restatement of timm 1.0.24 ``vision_transformer.Block`` -- so goldens
pin everything that lives under /root/reference, while the ViT block itself stays
"parity unpinned" (cross-checked separately against transformers' Dinov2Layer).

Never imported by the product, the tests or the bench; only by tools/make_goldens.py.
"""

from __future__ import annotations

import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"

# Queue of explicit DropPath masks ([B] 0/1 tensors) consumed in call order; when
# empty DropPath draws from torch's RNG like timm does.
DROP_PATH_MASKS: list[torch.Tensor] = []


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True) -> None:
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        if DROP_PATH_MASKS:
            m = DROP_PATH_MASKS.pop(0).to(x.dtype)
        else:
            m = x.new_empty(x.shape[0]).bernoulli_(keep)
        return x * (m / keep).view(-1, *([1] * (x.dim() - 1)))


def install() -> None:
    sys.path.insert(0, "/root/repo")
    from oracle import encoder as oenc

    class Block(nn.Module):
        """timm-signature adapter around the oracle's restated block."""

        def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, proj_drop=0.0,
                     attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, init_values=None,
                     **_kw) -> None:
            super().__init__()
            assert qkv_bias and proj_drop == 0.0 and attn_drop == 0.0
            self.norm1 = norm_layer(dim)
            self.attn = oenc._Attention(dim, num_heads)
            self.ls1 = oenc._LayerScale(dim, init_values)
            self.drop_path1 = DropPath(drop_path) if drop_path > 0 else nn.Identity()
            self.norm2 = norm_layer(dim)
            self.mlp = oenc._Mlp(dim, int(dim * mlp_ratio))
            self.ls2 = oenc._LayerScale(dim, init_values)
            self.drop_path2 = DropPath(drop_path) if drop_path > 0 else nn.Identity()

        def forward(self, x):
            x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
            return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_vit = types.ModuleType("timm.models.vision_transformer")
    timm_layers = types.ModuleType("timm.layers")
    timm_vit.Block = Block
    timm_layers.DropPath = DropPath
    timm_layers.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    timm_layers.trunc_normal_ = lambda t, std=1.0, **kw: nn.init.trunc_normal_(t, 0.0, std, -2, 2)
    timm.models, timm.layers = timm_models, timm_layers
    timm_models.vision_transformer = timm_vit
    for name, mod in [("timm", timm), ("timm.models", timm_models),
                      ("timm.models.vision_transformer", timm_vit), ("timm.layers", timm_layers)]:
        sys.modules[name] = mod

    kornia = types.ModuleType("kornia")
    kaug = types.ModuleType("kornia.augmentation")

    class _Dummy(nn.Module):
        def __init__(self, *a, **k) -> None:
            super().__init__()

    for n in ["AugmentationSequential", "RandomResizedCrop", "RandomHorizontalFlip",
              "RandomVerticalFlip", "RandomRotation90"]:
        setattr(kaug, n, _Dummy)
    kornia.augmentation = kaug
    sys.modules["kornia"] = kornia
    sys.modules["kornia.augmentation"] = kaug
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE_ROOT)
