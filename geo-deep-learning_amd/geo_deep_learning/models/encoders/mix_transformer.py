"""SegFormer MixVisionTransformer (MiT-B0..B5) on MI355X.

Drop-in for the reference's models/encoders/mix_transformer.py (same class names, constructor
arguments, ``get_encoder`` factory and state-dict keys); arithmetic in HIP kernels:

* OverlapPatchEmbed: 7x7/s4 stem on the raw bands as a 3x3 convolution on the 4x4 space-to-depth image (3..10 input
  channels are too few for the implicit-GEMM kernel), 3x3/s2 stages 2-4 as implicit-GEMM convs
  straight from the previous stage's NHWC feature; LayerNorm (eps 1e-5) -> f32 token stream.
* Attention (spatial reduction): q GEMM; K/V from ``LN(sr-conv(x))`` (k=stride=sr implicit-GEMM
  conv) -> kv GEMM; fused flash attention with Nq != Nkv reading Q and K in place (strided), V
  re-laid as V^T; proj GEMM with DropPath scale + residual fused.
* Mix-FFN: fc1 GEMM -> depthwise 3x3 + bias + erf-GELU in ONE HBM-bound kernel -> fc2 GEMM with
  DropPath scale + residual fused.

Every block is ONE autograd node (gdlhip.tnn._MitBlock): its backward recomputes the attention
probabilities, runs the dS/dQ GEMMs and batched dK/dV weight-gradient kernels, the depthwise /
LayerNorm / bias reductions, and fuses the residual-stream gradient adds into the LayerNorm-backward
kernel -- so ``loss.backward()`` trains the encoder with HIP kernels only.
"""

from __future__ import annotations

from functools import partial

import torch
from torch import Tensor, nn

from geo_deep_learning.models.segmentation.base import EncoderMixin
from geo_deep_learning.models.utils import _cl_conv
from gdlhip import nn as gnn
from gdlhip import cnn, ops, tnn


def _drop_scale(prob: float, training: bool, batch: int, device, mask: Tensor | None) -> Tensor | None:
    if prob == 0.0 or not training:
        return None
    keep = 1.0 - prob
    if mask is None:
        mask = torch.empty(batch, device=device, dtype=torch.float32).bernoulli_(keep)
    return (mask.to(device=device, dtype=torch.float32) / keep).contiguous()


class DWConv(nn.Module):
    """Depthwise 3x3 conv parameter container (mix_transformer.py:533-546)."""

    def __init__(self, dim: int = 768) -> None:
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def taps(self) -> Tensor:
        """[9, C] f32 tap-major weights (cached repack of the [C,1,3,3] parameter)."""
        w = self.dwconv.weight
        return gnn.cached((w,), "dw9", lambda: w.detach().reshape(w.shape[0], 9).t().contiguous())


class Mlp(nn.Module):
    """Mix-FFN (mix_transformer.py:17-63)."""

    def __init__(self, in_features: int, hidden_features: int | None = None, out_features: int | None = None,
                 act_layer: nn.Module = nn.GELU, drop: float = 0.0) -> None:
        super().__init__()
        if drop or act_layer is not nn.GELU:
            msg = "gdlhip Mix-FFN: GELU, drop=0 (every MiT variant)"
            raise NotImplementedError(msg)
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)


class Attention(nn.Module):
    """Spatial-reduction attention (mix_transformer.py:66-157)."""

    def __init__(self, dim: int, num_heads: int = 8, qk_scale: float | None = None, attn_drop: float = 0.0,
                 proj_drop: float = 0.0, sr_ratio: int = 1, *, qkv_bias: bool = False) -> None:
        super().__init__()
        if dim % num_heads != 0:
            msg = f"dim {dim} should be divided by num_heads {num_heads}."
            raise ValueError(msg)
        if qk_scale is not None or attn_drop or proj_drop or not qkv_bias:
            msg = "gdlhip MiT Attention: qkv_bias=True, default scale, no dropout (every MiT variant)"
            raise NotImplementedError(msg)
        self.dim = dim
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = _cl_conv(dim, dim, sr_ratio, padding=0, bias=True)
            self.sr.stride = (sr_ratio, sr_ratio)
            self.norm = nn.LayerNorm(dim)


class Block(nn.Module):
    """mix_transformer.py:160-221."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qk_scale: float | None = None,
                 drop: float = 0.0, attn_drop: float = 0.0, drop_path: float = 0.0, act_layer: nn.Module = nn.GELU,
                 norm_layer: nn.Module = nn.LayerNorm, sr_ratio: int = 1, *, qkv_bias: bool = False) -> None:
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, sr_ratio=sr_ratio)
        self.drop_prob = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _params(self) -> tuple:
        at, m = self.attn, self.mlp
        prm = [self.norm1.weight, self.norm1.bias, at.q.weight, at.q.bias, at.kv.weight, at.kv.bias, at.proj.weight,
               at.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.dwconv.dwconv.weight,
               m.dwconv.dwconv.bias, m.fc2.weight, m.fc2.bias]
        if at.sr_ratio > 1:
            prm += [at.sr.weight, at.sr.bias, at.norm.weight, at.norm.bias]
        return tuple(prm)

    def forward(self, x: Tensor, h: int, w: int, masks=None, scales=None) -> Tensor:
        """x: f32 token stream [B, N, C]; ``masks`` pins the two DropPath draws (tests); ``scales`` = the two per-sample
        scales already drawn for this block (gdlhip.nn.drop_path_scales: one launch for the whole encoder)."""
        b = x.shape[0]
        if scales is not None and masks is None:
            s1, s2 = scales
        else:
            s1 = _drop_scale(self.drop_prob, self.training, b, x.device, None if masks is None else masks[0])
            s2 = _drop_scale(self.drop_prob, self.training, b, x.device, None if masks is None else masks[1])
        at = self.attn
        eps_sr = at.norm.eps if at.sr_ratio > 1 else 0.0
        return tnn.mit_block(x, s1, s2, h, w, at.num_heads, at.sr_ratio, self.norm1.eps, eps_sr, gnn.compute_dtype(),
                             self._params())


class OverlapPatchEmbed(nn.Module):
    """Overlapped patch embedding: conv(k, stride, pad k//2) + LayerNorm (mix_transformer.py:224-276)."""

    def __init__(self, img_size: int = 224, patch_size: int = 7, stride: int = 4, in_chans: int = 3,
                 embed_dim: int = 768) -> None:
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.stride = stride
        self.is_stem = in_chans < 32   # raw bands (NCHW image) vs an NHWC feature of the previous stage
        self.proj = _cl_conv(in_chans, embed_dim, patch_size, padding=patch_size // 2, bias=True)
        self.proj.stride = (stride, stride)
        self.norm = nn.LayerNorm(embed_dim)

    def _stem_weight(self, cd: torch.dtype, kpad: int) -> Tensor:
        """[N, Kpad] GEMM operand of the image stem, k = (c, r, s) like the patchify kernel."""
        w = self.proj.weight

        def build():
            n = w.shape[0]
            flat = w.detach().reshape(n, -1)  # logical OIHW flatten -> (c, r, s)
            out = torch.zeros((n, kpad), device=w.device, dtype=torch.float32)
            out[:, : flat.shape[1]] = flat
            return out if cd == torch.float32 else ops.cast(out, cd)
        return gnn.cached((w,), f"stem:{cd}:{kpad}", build)

    def forward(self, x: Tensor) -> tuple[Tensor, int, int]:
        """NCHW f32 image (stage 1) or NHWC feature in the compute dtype (stages 2-4) ->
        (f32 tokens [B, h*w, C], h, w)."""
        cd = gnn.compute_dtype()
        k, s, p = self.patch_size[0], self.stride, self.patch_size[0] // 2
        n = self.proj.weight.shape[0]
        if self.is_stem and s in (2, 4) and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0 and not cnn.STEM_IM2COL:
            # raw bands, NCHW, fewer channels than one K chunk: no im2col matrix either -- the image is re-laid into 4 x 4
            # pixel blocks (16 C channels) and the k x k / s convolution runs as 3x3 phase convolutions on that map
            b, c, hi, wi = x.shape
            h, w = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
            cnn.mark_stem(self.proj.weight, s, p)
            xs = cnn.space_to_depth_image(ops.image_f32(x, "OverlapPatchEmbed"), cd)
            y = tnn.stem_conv(xs, self.proj.weight, self.proj.bias, torch.float32)
        elif self.is_stem:      # odd strides / image sizes: strided patchify (an im2col matrix) + GEMM
            b, c, hi, wi = x.shape
            h, w = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
            bke = 32 if cd == torch.float32 else 64
            kpad = (c * k * k + bke - 1) // bke * bke
            cols = ops.patchify(ops.image_f32(x, "OverlapPatchEmbed"), k, p, h, w, kpad, cd, stride=s)
            y = tnn.stem_linear(cols, self.proj.weight, self.proj.bias, self._stem_weight(cd, kpad), torch.float32)
        else:
            b, hi, wi, c = x.shape  # NHWC feature of the previous stage
            y = tnn.conv(x, self.proj.weight, self.proj.bias, stride=s, pad=p, out_dtype=torch.float32)
            h, w = y.shape[1], y.shape[2]
        return tnn.layernorm(y.view(b, h * w, n), self.norm, torch.float32), h, w


class MixVisionTransformer(nn.Module):
    """mix_transformer.py:279-530."""

    def __init__(self, img_size: int = 224, in_chans: int = 3, num_classes: int = 1000,
                 embed_dims: list[int] | None = None, num_heads: list[int] | None = None,
                 mlp_ratios: list[float] | None = None, qk_scale: float | None = None, drop_rate: float = 0.0,
                 attn_drop_rate: float = 0.0, drop_path_rate: float = 0.0, norm_layer: nn.Module = nn.LayerNorm,
                 depths: list[int] | None = None, sr_ratios: list[int] | None = None, *,
                 qkv_bias: bool = False) -> None:
        super().__init__()
        self.num_classes = num_classes
        embed_dims = embed_dims or [64, 128, 256, 512]
        num_heads = num_heads or [1, 2, 4, 8]
        mlp_ratios = mlp_ratios or [4, 4, 4, 4]
        depths = depths or [3, 4, 6, 3]
        sr_ratios = sr_ratios or [8, 4, 2, 1]
        self.depths = depths
        self.embed_dims = embed_dims
        for i in range(4):
            setattr(self, f"patch_embed{i + 1}", OverlapPatchEmbed(
                img_size=img_size // (1 if i == 0 else 2 ** (i + 1)), patch_size=7 if i == 0 else 3,
                stride=4 if i == 0 else 2, in_chans=in_chans if i == 0 else embed_dims[i - 1],
                embed_dim=embed_dims[i]))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(4):
            setattr(self, f"block{i + 1}", nn.ModuleList([
                Block(dim=embed_dims[i], num_heads=num_heads[i], mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias,
                      qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[cur + j],
                      norm_layer=norm_layer, sr_ratio=sr_ratios[i]) for j in range(depths[i])]))
            setattr(self, f"norm{i + 1}", norm_layer(embed_dims[i]))
            cur += depths[i]

    def _embed(self, i: int) -> nn.Module:
        return getattr(self, f"patch_embed{i + 1}")

    def forward_features_nhwc(self, x: Tensor, drop_masks=None) -> list[Tensor]:
        """-> 4 NHWC features in the compute dtype ([B,128,128,64] ... [B,16,16,512] for B2 @512^2)."""
        cd = gnn.compute_dtype()
        b = x.shape[0]
        outs, bi = [], 0
        scales = None
        if self.training and drop_masks is None:      # every DropPath draw of the pass in one launch
            blocks = [blk for i in range(4) for blk in getattr(self, f"block{i + 1}")]
            scales = gnn.drop_path_scales([blk.drop_prob for blk in blocks], b, x.device)
        for i in range(4):
            tok, h, w = self._embed(i)(x)
            for blk in getattr(self, f"block{i + 1}"):
                tok = blk(tok, h, w, None if drop_masks is None else drop_masks[bi], None if scales is None else scales[bi])
                bi += 1
            x = tnn.layernorm(tok, getattr(self, f"norm{i + 1}"), cd).view(b, h, w, -1)
            outs.append(x)
        return outs

    def forward_features(self, x: Tensor) -> list[Tensor]:
        return [ops.as_nchw(f) for f in self.forward_features_nhwc(x)]

    def forward(self, x: Tensor) -> list[Tensor]:
        return self.forward_features(x)


class MixVisionTransformerEncoder(MixVisionTransformer, EncoderMixin):
    """mix_transformer.py:550-584."""

    def __init__(self, in_channels: int, out_channels: int, depth: int = 5, **kwargs: object) -> None:
        super().__init__(in_chans=in_channels, **kwargs)
        self._out_channels = out_channels
        self._depth = depth

    def make_dilated(self) -> None:
        msg = "MixVisionTransformer encoder does not support dilated mode"
        raise ValueError(msg)

    def set_in_channels(self, in_channels: int) -> None:  # noqa: ARG002 (reference: a no-op, :570-574)
        return

    def forward(self, x: Tensor) -> list[Tensor]:
        return self.forward_features(x)[: self._depth - 1]

    def forward_nhwc(self, x: Tensor, drop_masks=None) -> list[Tensor]:
        return self.forward_features_nhwc(x, drop_masks)[: self._depth - 1]

    def load_state_dict(self, state_dict, *args, **kwargs):
        state_dict = dict(state_dict)
        state_dict.pop("head.weight", None)
        state_dict.pop("head.bias", None)
        return super().load_state_dict(state_dict, *args, **kwargs)


def _variant(embed_dims, depths):
    return {"encoder": MixVisionTransformerEncoder,
            "params": {"out_channels": (3, 0, *embed_dims), "embed_dims": embed_dims, "num_heads": [1, 2, 5, 8],
                       "mlp_ratios": [4, 4, 4, 4], "qkv_bias": True, "norm_layer": partial(nn.LayerNorm, eps=1e-6),
                       "depths": depths, "sr_ratios": [8, 4, 2, 1], "drop_rate": 0.0, "drop_path_rate": 0.1}}


# mix_transformer.py:599-708
mix_transformer_encoders = {
    "mit_b0": _variant([32, 64, 160, 256], [2, 2, 2, 2]),
    "mit_b1": _variant([64, 128, 320, 512], [2, 2, 2, 2]),
    "mit_b2": _variant([64, 128, 320, 512], [3, 4, 6, 3]),
    "mit_b3": _variant([64, 128, 320, 512], [3, 4, 18, 3]),
    "mit_b4": _variant([64, 128, 320, 512], [3, 8, 27, 3]),
    "mit_b5": _variant([64, 128, 320, 512], [3, 6, 40, 3]),
}


PRETRAINED_SETTINGS = ("imagenet",)      # mix_transformer.py:587-596: one published weight set per variant


def pretrained_checkpoint_path(name: str) -> tuple[str | None, list[str]]:
    """Where the ImageNet weights of ``name`` are looked for.  The reference calls ``model_zoo.load_url`` on
    ``github.com/qubvel/segmentation_models.pytorch/releases/download/v0.0.2/<name>.pth`` (mix_transformer.py:587-596,746),
    which caches the file as ``<torch hub dir>/checkpoints/<name>.pth``.  This build never opens a network connection: it reads
    that same cache location (or the file ``$GDL_MIT_CHECKPOINT`` names; a directory there is searched for ``<name>.pth``)."""
    import os
    from pathlib import Path
    env = os.environ.get("GDL_MIT_CHECKPOINT")
    candidates = []
    if env:
        candidates.append(str(Path(env) / f"{name}.pth") if Path(env).is_dir() else env)
    candidates.append(str(Path(torch.hub.get_dir()) / "checkpoints" / f"{name}.pth"))
    for path in candidates:
        if Path(path).is_file():
            return path, candidates
    return None, candidates


def get_encoder(name: str, in_channels: int = 3, depth: int = 5, weights: str | None = None,
                output_stride: int = 32) -> MixVisionTransformerEncoder:
    """mix_transformer.py:711-759.  ``weights="imagenet"`` (what configs/segformer_config_RGB.yaml:46 asks for) loads the
    published checkpoint from torch-hub's cache location without downloading (see ``pretrained_checkpoint_path``); like the
    reference, pretrained weights are only defined for 3-band input (a warning otherwise) and the classifier ``head.*`` entries
    of the checkpoint are dropped (:580-585)."""
    import warnings
    try:
        entry = mix_transformer_encoders[name]
    except KeyError as err:
        msg = f"Wrong encoder name `{name}`, supported encoders: {list(mix_transformer_encoders.keys())}"
        raise KeyError(msg) from err
    params = dict(entry["params"])
    params.update(in_channels=in_channels, depth=depth)
    encoder = entry["encoder"](**params)
    if weights is not None:
        if in_channels == 3:
            if weights not in PRETRAINED_SETTINGS:
                msg = (f"Wrong pretrained weights `{weights}` for encoder `{name}`. Available options are: "
                       f"{list(PRETRAINED_SETTINGS)}")
                raise KeyError(msg)
            path, candidates = pretrained_checkpoint_path(name)
            if path is None:
                msg = (f"weights='{weights}' needs the published checkpoint {name}.pth "
                       f"(https://github.com/qubvel/segmentation_models.pytorch/releases/download/v0.0.2/{name}.pth); this build "
                       f"does not download: put it at {candidates[-1]} or point GDL_MIT_CHECKPOINT at it, or pass weights=None")
                raise RuntimeError(msg)
            encoder.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
        else:
            warnings.warn("MixVisionTransformer encoder does not support pretrained weights for non-RGB input channels",
                          stacklevel=2)
    encoder.set_in_channels(in_channels)
    if output_stride != 32:
        encoder.make_dilated()           # raises ValueError like the reference (:565-568,755-757)
    return encoder


class DynamicChannelEmbed(nn.Module):
    """Channel-adaptive stem (mix_transformer.py:762-859): every band passes ONE shared 7x7/stride-4 convolution, is
    scaled by a weight vector generated from its sinusoidal position code, and the bands are pooled per pixel by a
    softmax attention over bands; then Linear + LayerNorm.  Any band count up to 16.

    The shared conv is the patchify + GEMM path of the image stem with B*C single-band images; the band weighting /
    attention / pooling is one fused HIP kernel pair (csrc/dynembed.hip)."""

    def __init__(self, patch_size: int = 7, stride: int = 4, embed_dim: int = 64, hidden_dim: int = 128) -> None:
        super().__init__()
        self.patch_size = patch_size
        self.stride = stride
        self.embed_dim = embed_dim
        self.pos_dim = hidden_dim
        self.weight_gen = nn.Sequential(nn.Linear(self.pos_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, embed_dim),
                                        nn.Tanh())
        self.spatial_conv = nn.Conv2d(1, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.channel_attention = nn.Sequential(nn.Conv1d(embed_dim + self.pos_dim, embed_dim // 2, 1), nn.ReLU(),
                                               nn.Conv1d(embed_dim // 2, 1, 1))
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.norm = nn.LayerNorm(embed_dim)
        self._pos: dict = {}

    def get_position_encoding(self, n_channels: int, device: torch.device) -> Tensor:
        """Sinusoidal band codes [C, pos_dim] (:810-821); a constant per band count, built once on the host."""
        key = (n_channels, str(device))
        if key not in self._pos:
            positions = torch.arange(n_channels).float()
            dim_t = torch.arange(0, self.pos_dim, 2).float()
            inv_freq = 1.0 / (10000 ** (dim_t / self.pos_dim))
            pe = torch.zeros(n_channels, self.pos_dim)
            pe[:, 0::2] = torch.sin(positions.unsqueeze(1) * inv_freq)
            pe[:, 1::2] = torch.cos(positions.unsqueeze(1) * inv_freq)
            self._pos[key] = pe.to(device)
        return self._pos[key]

    def _stem_weight(self, cd: torch.dtype, kpad: int) -> Tensor:
        w = self.spatial_conv.weight

        def build():
            flat = w.detach().reshape(w.shape[0], -1)
            out = torch.zeros((w.shape[0], kpad), device=w.device, dtype=torch.float32)
            out[:, : flat.shape[1]] = flat
            return out if cd == torch.float32 else ops.cast(out, cd)
        return gnn.cached((w,), f"dynstem:{cd}:{kpad}", build)

    def forward(self, x: Tensor) -> tuple[Tensor, int, int]:
        """NCHW f32 image with any number of bands -> (f32 tokens [B, h*w, E], h, w)."""
        cd = gnn.compute_dtype()
        b, c, hi, wi = x.shape
        k, s, p = self.patch_size, self.stride, self.patch_size // 2
        h, w = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
        planes = ops.image_f32(x, "DynamicChannelEmbed").view(b * c, 1, hi, wi)       # every band is its own one-channel image
        if s in (2, 4) and hi % 4 == 0 and wi % 4 == 0 and not cnn.STEM_IM2COL:
            cnn.mark_stem(self.spatial_conv.weight, s, p)                              # im2col-free, like OverlapPatchEmbed
            conv = tnn.stem_conv(cnn.space_to_depth_image(planes, cd), self.spatial_conv.weight, self.spatial_conv.bias,
                                 torch.float32)
        else:
            bke = 32 if cd == torch.float32 else 64
            kpad = (k * k + bke - 1) // bke * bke
            cols = ops.patchify(planes, k, p, h, w, kpad, cd, stride=s)
            conv = tnn.stem_linear(cols, self.spatial_conv.weight, self.spatial_conv.bias, self._stem_weight(cd, kpad),
                                   torch.float32)
        agg = tnn.chan_pool(conv.view(b, c, h * w, self.embed_dim), self.get_position_encoding(c, x.device),
                            self.weight_gen, self.channel_attention)
        tok = gnn.to_compute(agg.view(b, h, w, self.embed_dim), cd)
        y = tnn.linear(tok, self.proj.weight, self.proj.bias, out_dtype=torch.float32)
        return tnn.layernorm(y.view(b, h * w, self.embed_dim), self.norm, torch.float32), h, w


class DynamicMixTransformer(nn.Module):
    """MiT whose first patch embedding is the channel-adaptive stem (mix_transformer.py:862-934); stages 2-4, the blocks
    and the norms are those of the named variant."""

    def __init__(self, encoder: str = "mit_b0", in_channels: int = 3, weights: str | None = None) -> None:
        super().__init__()
        base = get_encoder(name=encoder, in_channels=in_channels, weights=weights)
        self.dynamic_patch_embed1 = DynamicChannelEmbed(patch_size=7, stride=4,
                                                        embed_dim=base.patch_embed1.proj.weight.shape[0], hidden_dim=128)
        for i in (2, 3, 4):
            setattr(self, f"patch_embed{i}", getattr(base, f"patch_embed{i}"))
        for i in (1, 2, 3, 4):
            setattr(self, f"block{i}", getattr(base, f"block{i}"))
        for i in (1, 2, 3, 4):
            setattr(self, f"norm{i}", getattr(base, f"norm{i}"))
        self.depths = base.depths

    def _embed(self, i: int) -> nn.Module:
        return self.dynamic_patch_embed1 if i == 0 else getattr(self, f"patch_embed{i + 1}")

    forward_features_nhwc = MixVisionTransformer.forward_features_nhwc

    def forward_nhwc(self, x: Tensor, drop_masks=None) -> list[Tensor]:
        return self.forward_features_nhwc(x, drop_masks)

    def forward_features(self, x: Tensor) -> list[Tensor]:
        return [ops.as_nchw(f) for f in self.forward_features_nhwc(x)]

    def forward(self, x: Tensor) -> list[Tensor]:
        return self.forward_features(x)
