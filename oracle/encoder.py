"""Oracle restatement of the DOFA-v2 encoder (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/geo_deep_learning/models/encoders/dofa_v2.py and, for
the ViT block, timm 1.0.24 ``vision_transformer.Block`` (third-party, not under
/root/reference: restated from its published algorithm).
State-dict keys are identical to the reference's so the same weights load.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor, nn


def position_embedding(embed_dim: int, pos: Tensor) -> Tensor:
    """1-D sin/cos embedding.  Reference: dofa_v2.py:9-35."""
    omega = torch.arange(embed_dim // 2, dtype=torch.float32, device=pos.device)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000**omega
    out = torch.einsum("m,d->md", pos.reshape(-1), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


class FCResLayer(nn.Module):
    """x + relu(w2(relu(w1 x))).  Reference: dofa_v2.py:38-56."""

    def __init__(self, linear_size: int = 128) -> None:
        super().__init__()
        self.w1 = nn.Linear(linear_size, linear_size)
        self.w2 = nn.Linear(linear_size, linear_size)

    def forward(self, x: Tensor) -> Tensor:
        return x + F.relu(self.w2(F.relu(self.w1(x))))


class _SelfAttn(nn.Module):
    """nn.MultiheadAttention parameters (packed in_proj), unbatched (S,E) input."""

    def __init__(self, dim: int, heads: int) -> None:
        super().__init__()
        self.heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = nn.Linear(dim, dim)

    def forward(self, x: Tensor) -> Tensor:
        s, e = x.shape
        hd = e // self.heads
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)
        q, k, v = qkv.split(e, dim=1)
        q = q.reshape(s, self.heads, hd).transpose(0, 1)
        k = k.reshape(s, self.heads, hd).transpose(0, 1)
        v = v.reshape(s, self.heads, hd).transpose(0, 1)
        att = torch.softmax((q @ k.transpose(1, 2)) / math.sqrt(hd), dim=-1)
        o = (att @ v).transpose(0, 1).reshape(s, e)
        return self.out_proj(o)


class _PostNormEncoderLayer(nn.Module):
    """nn.TransformerEncoderLayer(norm_first=False, gelu, dropout 0, ffn 2048).

    Reference call site: dofa_v2.py:73-85 (torch defaults: SURVEY App. A.2).
    """

    def __init__(self, dim: int, heads: int, ffn: int = 2048) -> None:
        super().__init__()
        self.self_attn = _SelfAttn(dim, heads)
        self.linear1 = nn.Linear(dim, ffn)
        self.linear2 = nn.Linear(ffn, dim)
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)

    def forward(self, x: Tensor) -> Tensor:
        x = self.norm1(x + self.self_attn(x))
        return self.norm2(x + self.linear2(F.gelu(self.linear1(x))))


class _Encoder(nn.Module):
    def __init__(self, dim: int, heads: int, num_layers: int) -> None:
        super().__init__()
        self.layers = nn.ModuleList(
            [_PostNormEncoderLayer(dim, heads) for _ in range(num_layers)]
        )

    def forward(self, x: Tensor) -> Tensor:
        for layer in self.layers:
            x = layer(x)
        return x


class TransformerWeightGenerator(nn.Module):
    """Reference: dofa_v2.py:59-106."""

    def __init__(self, input_dim: int, output_dim: int, embed_dim: int,
                 num_heads: int = 4, num_layers: int = 1) -> None:
        super().__init__()
        self.transformer_encoder = _Encoder(input_dim, num_heads, num_layers)
        self.fc_weight = nn.Linear(input_dim, output_dim)
        self.fc_bias = nn.Linear(input_dim, embed_dim)
        self.wt_num = 128
        self.weight_tokens = nn.Parameter(torch.empty([self.wt_num, input_dim]))
        self.bias_token = nn.Parameter(torch.empty([1, input_dim]))
        nn.init.normal_(self.weight_tokens, std=0.02)
        nn.init.normal_(self.bias_token, std=0.02)

    def forward(self, x: Tensor) -> tuple[Tensor, Tensor]:
        pos_wave = x
        x = torch.cat([self.weight_tokens, pos_wave, self.bias_token], dim=0)
        out = self.transformer_encoder(x)
        weights = self.fc_weight(out[self.wt_num:-1] + pos_wave)
        bias = self.fc_bias(out[-1])
        return weights, bias


class DOFAv2Embedding(nn.Module):
    """Wavelength-conditioned dynamic patch embed.  Reference: dofa_v2.py:109-181."""

    def __init__(self, dynamic_embed_dim: int = 128, kernel_size: int = 14,
                 embed_dim: int = 768) -> None:
        super().__init__()
        self.dynamic_embed_dim = dynamic_embed_dim
        self.kernel_size = kernel_size
        self.embed_dim = embed_dim
        self.weight_generator = TransformerWeightGenerator(
            dynamic_embed_dim, kernel_size * kernel_size * embed_dim, embed_dim)
        self.fclayer = FCResLayer(dynamic_embed_dim)
        self.scaler = 0.01

    def dynamic_kernel(self, wavelengths: Tensor) -> tuple[Tensor, Tensor]:
        """-> conv weight [D, C, k, k] and bias [D] (dofa_v2.py:152-166)."""
        waves = position_embedding(self.dynamic_embed_dim, wavelengths * 1000)
        waves = self.fclayer(waves)
        weight, bias = self.weight_generator(waves)
        c = wavelengths.numel()
        k = self.kernel_size
        w = weight.view(c, k, k, self.embed_dim).permute(3, 0, 1, 2) * self.scaler
        b = bias.view(self.embed_dim) * self.scaler
        return w, b

    def forward(self, x: Tensor, wavelengths: Tensor) -> Tensor:
        w, b = self.dynamic_kernel(wavelengths)
        x = F.conv2d(x, w, bias=b, stride=self.kernel_size, padding=1)
        return x.flatten(2).transpose(1, 2)


class _Attention(nn.Module):
    """timm Attention (qkv_bias=True, no q/k norm)."""

    def __init__(self, dim: int, num_heads: int) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: Tensor) -> Tensor:
        b, n, c = x.shape
        hd = c // self.num_heads
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        att = torch.softmax((q * hd**-0.5) @ k.transpose(-2, -1), dim=-1)
        x = (att @ v).transpose(1, 2).reshape(b, n, c)
        return self.proj(x)


class _LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float) -> None:
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x: Tensor) -> Tensor:
        return x * self.gamma


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: Tensor) -> Tensor:
        return self.fc2(F.gelu(self.fc1(x)))


def drop_path(x: Tensor, p: float, training: bool, mask: Tensor | None) -> Tensor:
    """timm DropPath: per-sample Bernoulli(1-p)/(1-p).  An explicit ``mask``
    (shape [B], values 0/1) pins the random draw for parity tests."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    if mask is None:
        mask = x.new_empty(x.shape[0]).bernoulli_(keep)
    return x * (mask.to(x.dtype) / keep).view(-1, *([1] * (x.dim() - 1)))


class Block(nn.Module):
    """timm 1.0.24 ViT Block as built at dofa_v2.py:250-260 (LN eps 1e-5)."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0,
                 drop_path: float = 0.0, init_values: float = 1e-5) -> None:
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attention(dim, num_heads)
        self.ls1 = _LayerScale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = _LayerScale(dim, init_values)
        self.drop_prob = float(drop_path)

    def forward(self, x: Tensor, masks: tuple[Tensor, Tensor] | None = None) -> Tensor:
        m1, m2 = masks if masks is not None else (None, None)
        x = x + drop_path(self.ls1(self.attn(self.norm1(x))), self.drop_prob, self.training, m1)
        return x + drop_path(self.ls2(self.mlp(self.norm2(x))), self.drop_prob, self.training, m2)


def get_1d_sincos(embed_dim: int, pos: Tensor) -> Tensor:
    """Reference: dofa_v2.py:420-433."""
    omega = torch.arange(embed_dim // 2, dtype=torch.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000**omega
    out = torch.einsum("m,d->md", pos.reshape(-1), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int, *, cls_token: bool) -> Tensor:
    """Reference: dofa_v2.py:394-418."""
    grid = torch.meshgrid(torch.arange(grid_size), torch.arange(grid_size), indexing="ij")
    grid = torch.stack(grid, dim=0).reshape([2, 1, grid_size, grid_size])
    emb = torch.cat([get_1d_sincos(embed_dim // 2, grid[0]),
                     get_1d_sincos(embed_dim // 2, grid[1])], dim=1)
    if cls_token:
        emb = torch.cat([torch.zeros([1, embed_dim]), emb], dim=0)
    return emb


class DOFAv2(nn.Module):
    """Reference: dofa_v2.py:184-501 (pretrained download path excluded)."""

    def __init__(self, img_size: tuple[int, int] = (224, 224), patch_size: int = 14,
                 embed_dim: int = 768, depth: int = 12, num_heads: int = 12,
                 mlp_ratio: float = 4.0, drop_path_rate: float = 0.1,
                 out_indices: list[int] | None = None, init_values: float = 1e-5) -> None:
        super().__init__()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.img_size = tuple(img_size)
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_patches = (img_size[0] // patch_size) * (img_size[1] // patch_size)
        self.out_indices = out_indices if out_indices is not None else [depth - 1]
        self.patch_embed = DOFAv2Embedding(128, patch_size, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim),
                                      requires_grad=False)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList(
            [Block(embed_dim, num_heads, mlp_ratio, dpr[i], init_values) for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)  # present in the state dict, never used (dofa_v2.py:478-486)
        self.pos_embed.data.copy_(get_2d_sincos_pos_embed(
            embed_dim, int(self.num_patches**0.5), cls_token=True).unsqueeze(0))
        nn.init.normal_(self.cls_token, std=0.02)

    def forward(self, x: Tensor, wavelengths: Tensor,
                drop_masks: list[tuple[Tensor, Tensor]] | None = None) -> list[Tensor]:
        """Reference: dofa_v2.py:435-487."""
        if wavelengths.dim() == 2:
            if not torch.allclose(wavelengths, wavelengths[0:1].expand_as(wavelengths)):
                msg = "DOFA cannot handle different wavelengths within a batch"
                raise ValueError(msg)
            wavelengths = wavelengths[0]
        x = self.patch_embed(x, wavelengths)
        x = x + self.pos_embed[:, 1:, :]
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        feats = []
        for i, blk in enumerate(self.blocks):
            x = blk(x, None if drop_masks is None else drop_masks[i])
            if i in self.out_indices:
                f = x[:, 1:, :]
                b, length, c = f.shape
                hw = int(length**0.5)
                feats.append(f.reshape(b, hw, hw, c).permute(0, 3, 1, 2))
        return feats


def create_dofa_base(img_size=(224, 224), **kw) -> DOFAv2:
    """Reference: dofa_v2.py:504-534."""
    return DOFAv2(img_size=img_size, patch_size=14, embed_dim=768, num_heads=12, depth=12,
                  out_indices=[4, 6, 10, 11], **kw)


def create_dofa_large(img_size=(224, 224), **kw) -> DOFAv2:
    """Reference: dofa_v2.py:537-567."""
    return DOFAv2(img_size=img_size, patch_size=14, embed_dim=1024, num_heads=16, depth=24,
                  out_indices=[5, 9, 15, 21], **kw)
