"""Oracle restatement of SegFormer: MixVisionTransformer encoder + all-MLP decoder.

TEST INFRASTRUCTURE ONLY.  Follows, with identical state-dict keys:
  /root/reference/geo_deep_learning/models/encoders/mix_transformer.py   (S1-S4, S7)
  /root/reference/geo_deep_learning/models/decoders/segformer_mlp.py     (S5)
  /root/reference/geo_deep_learning/models/segmentation/segformer.py     (S6)
Pinned against outputs of the real reference in tests/golden/segformer_*.npz.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .encoder import drop_path

# mix_transformer.py:599-708 (qkv_bias=True, LayerNorm eps 1e-6, drop_path_rate 0.1)
MIT_VARIANTS = {
    "mit_b0": dict(embed_dims=[32, 64, 160, 256], depths=[2, 2, 2, 2]),
    "mit_b1": dict(embed_dims=[64, 128, 320, 512], depths=[2, 2, 2, 2]),
    "mit_b2": dict(embed_dims=[64, 128, 320, 512], depths=[3, 4, 6, 3]),
    "mit_b3": dict(embed_dims=[64, 128, 320, 512], depths=[3, 4, 18, 3]),
    "mit_b4": dict(embed_dims=[64, 128, 320, 512], depths=[3, 8, 27, 3]),
    "mit_b5": dict(embed_dims=[64, 128, 320, 512], depths=[3, 6, 40, 3]),
}
NUM_HEADS = [1, 2, 5, 8]
SR_RATIOS = [8, 4, 2, 1]


class DWConv(nn.Module):
    """mix_transformer.py:533-546."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x: Tensor, h: int, w: int) -> Tensor:
        b, _, c = x.shape
        return self.dwconv(x.transpose(1, 2).view(b, c, h, w)).flatten(2).transpose(1, 2)


class Mlp(nn.Module):
    """Mix-FFN: fc1 -> depthwise 3x3 -> GELU -> fc2 (mix_transformer.py:17-63)."""

    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = DWConv(hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: Tensor, h: int, w: int) -> Tensor:
        return self.fc2(F.gelu(self.dwconv(self.fc1(x), h, w)))


class Attention(nn.Module):
    """Spatial-reduction attention (mix_transformer.py:66-157)."""

    def __init__(self, dim: int, num_heads: int, sr_ratio: int) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.kv = nn.Linear(dim, dim * 2, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)  # default eps 1e-5 (NOT the blocks' 1e-6)

    def forward(self, x: Tensor, h: int, w: int) -> Tensor:
        b, n, c = x.shape
        hd = c // self.num_heads
        q = self.q(x).reshape(b, n, self.num_heads, hd).permute(0, 2, 1, 3)
        if self.sr_ratio > 1:
            x_ = x.permute(0, 2, 1).reshape(b, c, h, w)
            x_ = self.sr(x_).reshape(b, c, -1).permute(0, 2, 1)
            x_ = self.norm(x_)
        else:
            x_ = x
        kv = self.kv(x_).reshape(b, -1, 2, self.num_heads, hd).permute(2, 0, 3, 1, 4)
        k, v = kv[0], kv[1]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(b, n, c))


class Block(nn.Module):
    """mix_transformer.py:160-221."""

    def __init__(self, dim: int, num_heads: int, sr_ratio: int, drop_path_rate: float, eps: float) -> None:
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, sr_ratio)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, dim * 4)
        self.drop_prob = float(drop_path_rate)

    def forward(self, x: Tensor, h: int, w: int, masks=None) -> Tensor:
        m1, m2 = masks if masks is not None else (None, None)
        x = x + drop_path(self.attn(self.norm1(x), h, w), self.drop_prob, self.training, m1)
        return x + drop_path(self.mlp(self.norm2(x), h, w), self.drop_prob, self.training, m2)


class OverlapPatchEmbed(nn.Module):
    """mix_transformer.py:224-276 (LayerNorm with the DEFAULT eps 1e-5)."""

    def __init__(self, patch_size: int, stride: int, in_chans: int, embed_dim: int) -> None:
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride,
                              padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x: Tensor):
        x = self.proj(x)
        _, _, h, w = x.shape
        return self.norm(x.flatten(2).transpose(1, 2)), h, w


class MixVisionTransformerEncoder(nn.Module):
    """mix_transformer.py:279-584; returns the 4 stage features (depth=5 -> [:4])."""

    def __init__(self, name: str = "mit_b2", in_channels: int = 3, drop_path_rate: float = 0.1) -> None:
        super().__init__()
        cfg = MIT_VARIANTS[name]
        dims, depths = cfg["embed_dims"], cfg["depths"]
        self.depths = depths
        eps = 1e-6
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(4):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(
                7 if i == 0 else 3, 4 if i == 0 else 2, in_channels if i == 0 else dims[i - 1], dims[i]))
            setattr(self, f"block{i + 1}", nn.ModuleList(
                [Block(dims[i], NUM_HEADS[i], SR_RATIOS[i], dpr[cur + j], eps) for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", nn.LayerNorm(dims[i], eps=eps))
            cur += depths[i]

    def _embed(self, i: int) -> nn.Module:
        return getattr(self, f"patch_embed{i + 1}")

    def forward(self, x: Tensor, drop_masks=None) -> list[Tensor]:
        b = x.shape[0]
        outs, bi = [], 0
        for i in range(4):
            x, h, w = self._embed(i)(x)
            for blk in getattr(self, f"block{i + 1}"):
                x = blk(x, h, w, None if drop_masks is None else drop_masks[bi])
                bi += 1
            x = getattr(self, f"norm{i + 1}")(x)
            x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()
            outs.append(x)
        return outs


class DynamicChannelEmbed(nn.Module):
    """mix_transformer.py:762-859 restated band by band.  With x_c the c-th band, k_c its sinusoidal code (:810-821):
        u_c   = spatial_conv(x_c)                        one shared 1->E 7x7/4 convolution         (:825-826)
        w_c   = tanh(L2 relu(L0 k_c))                    weight_gen                                 (:823, :781-786)
        v_c   = u_c * w_c                                                                           (:829-830)
        s_c   = A2 relu(A0 [v_c ; k_c])                  1x1 Conv1d pair on every (pixel, band)     (:832-846)
        out   = LayerNorm(proj(sum_c softmax_c(s)_c v_c))                                           (:847-853)
    """

    def __init__(self, patch_size: int = 7, stride: int = 4, embed_dim: int = 64, hidden_dim: int = 128) -> None:
        super().__init__()
        self.pos_dim = hidden_dim
        self.weight_gen = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, embed_dim),
                                        nn.Tanh())
        self.spatial_conv = nn.Conv2d(1, embed_dim, patch_size, stride=stride, padding=patch_size // 2)
        self.channel_attention = nn.Sequential(nn.Conv1d(embed_dim + hidden_dim, embed_dim // 2, 1), nn.ReLU(),
                                               nn.Conv1d(embed_dim // 2, 1, 1))
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.norm = nn.LayerNorm(embed_dim)

    def band_codes(self, n: int) -> Tensor:
        half = torch.arange(0, self.pos_dim, 2).float()
        ang = torch.arange(n).float()[:, None] / (10000 ** (half / self.pos_dim))[None, :]
        return torch.stack([ang.sin(), ang.cos()], dim=-1).flatten(1)      # sin on even, cos on odd columns

    def forward(self, x: Tensor):
        b, c, _, _ = x.shape
        k = self.band_codes(c).to(x)                                        # [C, D]
        u = torch.stack([self.spatial_conv(x[:, i:i + 1]) for i in range(c)], dim=1)   # [B, C, E, h, w]
        h, w = u.shape[-2:]
        v = u * self.weight_gen(k)[None, :, :, None, None]
        a0, a2 = self.channel_attention[0], self.channel_attention[2]
        feat = torch.cat([v, k[None, :, :, None, None].expand(b, c, -1, h, w)], dim=2)  # [B, C, E+D, h, w]
        hid = torch.einsum("bcfhw,jf->bcjhw", feat, a0.weight[:, :, 0]) + a0.bias[None, None, :, None, None]
        s = torch.einsum("bcjhw,j->bchw", hid.relu(), a2.weight[0, :, 0]) + a2.bias
        pooled = (v * s.softmax(dim=1)[:, :, None]).sum(dim=1)              # [B, E, h, w]
        tok = pooled.flatten(2).transpose(1, 2)
        return self.norm(self.proj(tok)), h, w


class DynamicMixTransformer(MixVisionTransformerEncoder):
    """mix_transformer.py:862-934: the named MiT with patch_embed1 replaced by DynamicChannelEmbed (state-dict key
    dynamic_patch_embed1.*, no patch_embed1.*)."""

    def __init__(self, name: str = "mit_b0", drop_path_rate: float = 0.1) -> None:
        super().__init__(name, 3, drop_path_rate)
        e = self.patch_embed1.proj.out_channels
        del self.patch_embed1
        self.dynamic_patch_embed1 = DynamicChannelEmbed(7, 4, e, 128)

    def _embed(self, i: int) -> nn.Module:
        return self.dynamic_patch_embed1 if i == 0 else getattr(self, f"patch_embed{i + 1}")


class MLP(nn.Module):
    """segformer_mlp.py:8-19."""

    def __init__(self, input_dim: int, embed_dim: int) -> None:
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward(self, x: Tensor) -> Tensor:
        return self.proj(x.flatten(2).transpose(1, 2))


class Decoder(nn.Module):
    """All-MLP decoder (segformer_mlp.py:22-130)."""

    def __init__(self, encoder: str = "mit_b2", num_classes: int = 1, dropout_ratio: float = 0.1) -> None:
        super().__init__()
        in_channels = MIT_VARIANTS[encoder]["embed_dims"]
        e = 256 if encoder in ("mit_b0", "mit_b1") else 768
        c1, c2, c3, c4 = in_channels
        self.linear_c4, self.linear_c3 = MLP(c4, e), MLP(c3, e)
        self.linear_c2, self.linear_c1 = MLP(c2, e), MLP(c1, e)
        self.linear_fuse = nn.Sequential(nn.Conv2d(e * 4, e, 1, bias=False), nn.BatchNorm2d(e), nn.ReLU())
        self.dropout_ratio = dropout_ratio
        self.linear_pred = nn.Conv2d(e, num_classes, kernel_size=1)

    def forward(self, feats, drop_mask: Tensor | None = None) -> Tensor:
        c1, c2, c3, c4 = feats
        n = c4.shape[0]
        size = c1.shape[2:]

        def lvl(lin, c, up=True):
            y = lin(c).permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
            return F.interpolate(y, size=size, mode="bilinear", align_corners=False) if up else y

        cat = torch.cat([lvl(self.linear_c4, c4), lvl(self.linear_c3, c3), lvl(self.linear_c2, c2),
                         lvl(self.linear_c1, c1, up=False)], dim=1)
        x = self.linear_fuse(cat)
        if self.training and self.dropout_ratio > 0:
            if drop_mask is None:
                drop_mask = x.new_empty(x.shape[:2]).bernoulli_(1 - self.dropout_ratio)
            x = x * (drop_mask.to(x.dtype) / (1 - self.dropout_ratio))[:, :, None, None]
        return self.linear_pred(x)


class SegFormerSegmentationModel(nn.Module):
    """models/segmentation/segformer.py:15-57 (weights=None)."""

    def __init__(self, encoder: str = "mit_b0", in_channels: int = 3, num_classes: int = 1, *,
                 use_dynamic_encoder: bool = False) -> None:
        super().__init__()
        self.encoder = DynamicMixTransformer(encoder) if use_dynamic_encoder else MixVisionTransformerEncoder(encoder, in_channels)
        self.decoder = Decoder(encoder, num_classes)

    def forward(self, img: Tensor, drop_masks=None, dec_drop_mask=None) -> Tensor:
        x = self.decoder(self.encoder(img, drop_masks), dec_drop_mask)
        return F.interpolate(x, size=img.shape[2:], mode="bilinear", align_corners=False)
