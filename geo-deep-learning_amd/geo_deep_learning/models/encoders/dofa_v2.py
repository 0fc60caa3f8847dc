"""DOFA-v2 encoder on MI355X (drop-in for the reference's models/encoders/dofa_v2.py).

Same public classes, constructor arguments and state-dict keys as the reference
(/root/reference/geo_deep_learning/models/encoders/dofa_v2.py); the arithmetic runs in
hand-written HIP kernels through ``gdlhip``:

* dynamic patch embed  = wavelength sincos embed -> FCRes -> 1 post-norm transformer layer ->
  fc_weight / fc_bias (all GEMMs on the MFMA kernel, exact-f32) -> im2col-free patch GEMM whose
  epilogue adds bias + pos_embed and writes the token stream in place;
* ViT block            = LayerNorm (wave-shuffle) -> qkv GEMM -> attention (flash kernel in bf16,
  materialised-score f32 path for parity) -> proj GEMM with LayerScale/DropPath/residual fused
  -> LayerNorm -> fc1 GEMM + erf-GELU -> fc2 GEMM with LayerScale/DropPath/residual fused.

The token stream ``[B, N, D]`` is kept in f32 (as under torch autocast); a feature tap
``x[:, 1:, :]`` is already an NHWC tensor, so ``reshape(B,H,W,C).permute(0,3,1,2)``
(dofa_v2.py:470-475) is returned as a zero-copy channels-last view.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from gdlhip import nn as gnn
from gdlhip import ops, tnn


def position_embedding(embed_dim: int, pos: Tensor) -> Tensor:
    """1-D sin/cos embedding (dofa_v2.py:9-35) -- HIP kernel."""
    if embed_dim % 2 != 0:
        msg = "embed_dim must be even"
        raise ValueError(msg)
    return ops.sincos_embed(pos, embed_dim)


class FCResLayer(nn.Module):
    """Fully-connected residual layer (dofa_v2.py:38-56)."""

    def __init__(self, linear_size: int = 128) -> None:
        super().__init__()
        self.l_size = linear_size
        self.w1 = nn.Linear(self.l_size, self.l_size)
        self.w2 = nn.Linear(self.l_size, self.l_size)

    def forward(self, x: Tensor) -> Tensor:
        f32 = torch.float32
        y = ops.linear(x, gnn.gemm_weight(self.w1.weight, f32), self.w1.bias.detach(), act=ops.ACT_RELU)
        return ops.linear(y, gnn.gemm_weight(self.w2.weight, f32), self.w2.bias.detach(),
                          act=ops.ACT_RELU, resid=x)


class _SelfAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's key names."""

    def __init__(self, dim: int, heads: int) -> None:
        super().__init__()
        self.num_heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = nn.Linear(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)


class _PostNormLayer(nn.Module):
    """nn.TransformerEncoderLayer(norm_first=False, activation=gelu, dropout=0) (dofa_v2.py:73-85)."""

    def __init__(self, dim: int, heads: int, ffn: int = 2048) -> None:
        super().__init__()
        self.self_attn = _SelfAttention(dim, heads)
        self.linear1 = nn.Linear(dim, ffn)
        self.linear2 = nn.Linear(ffn, dim)
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)

    def forward(self, x: Tensor) -> Tensor:
        f32 = torch.float32
        sa = self.self_attn
        qkv = ops.linear(x, gnn.gemm_weight(sa.in_proj_weight, f32), sa.in_proj_bias.detach())
        att = ops.attention_unfused(*ops.split_qkv(qkv.unsqueeze(0)), sa.num_heads)[0]
        h = ops.linear(att, gnn.gemm_weight(sa.out_proj.weight, f32), sa.out_proj.bias.detach(), resid=x)
        x = ops.layernorm(h, self.norm1.weight.detach(), self.norm1.bias.detach(), self.norm1.eps, f32)
        f = ops.linear(x, gnn.gemm_weight(self.linear1.weight, f32), self.linear1.bias.detach(),
                       act=ops.ACT_GELU)
        h = ops.linear(f, gnn.gemm_weight(self.linear2.weight, f32), self.linear2.bias.detach(), resid=x)
        return ops.layernorm(h, self.norm2.weight.detach(), self.norm2.bias.detach(), self.norm2.eps, f32)


class _EncoderStack(nn.Module):
    def __init__(self, dim: int, heads: int, num_layers: int) -> None:
        super().__init__()
        self.layers = nn.ModuleList([_PostNormLayer(dim, heads) for _ in range(num_layers)])

    def forward(self, x: Tensor) -> Tensor:
        for layer in self.layers:
            x = layer(x)
        return x


class TransformerWeightGenerator(nn.Module):
    """Dynamic weight generator (dofa_v2.py:59-106); runs in exact f32 (it is tiny)."""

    def __init__(self, input_dim: int, output_dim: int, embed_dim: int, num_heads: int = 4,
                 num_layers: int = 1) -> None:
        super().__init__()
        self.transformer_encoder = _EncoderStack(input_dim, num_heads, num_layers)
        self.fc_weight = nn.Linear(input_dim, output_dim)
        self.fc_bias = nn.Linear(input_dim, embed_dim)
        self.wt_num = 128
        self.weight_tokens = nn.Parameter(torch.empty([self.wt_num, input_dim]))
        self.bias_token = nn.Parameter(torch.empty([1, input_dim]))
        nn.init.normal_(self.weight_tokens, std=0.02)
        nn.init.normal_(self.bias_token, std=0.02)

    def forward(self, x: Tensor) -> tuple[Tensor, Tensor]:
        f32 = torch.float32
        pos_wave = x
        c, d = pos_wave.shape
        s = self.wt_num + c + 1
        seq = torch.empty((s, d), device=x.device, dtype=f32)
        ops.add_rows(self.weight_tokens.detach(), None, seq[: self.wt_num], self.wt_num)
        ops.add_rows(pos_wave, None, seq[self.wt_num: self.wt_num + c], c)
        ops.add_rows(self.bias_token.detach(), None, seq[s - 1:], 1)
        out = self.transformer_encoder(seq)
        tok = torch.empty((c, d), device=x.device, dtype=f32)
        ops.add_rows(out[self.wt_num: self.wt_num + c], pos_wave, tok, c)
        weights = ops.linear(tok, gnn.gemm_weight(self.fc_weight.weight, f32), self.fc_weight.bias.detach())
        bias = ops.linear(out[s - 1:], gnn.gemm_weight(self.fc_bias.weight, f32),
                          self.fc_bias.bias.detach())
        return weights, bias.reshape(-1)


class DOFAv2Embedding(nn.Module):
    """Dynamic One-For-All v2 embedding layer (dofa_v2.py:109-181)."""

    def __init__(self, dynamic_embed_dim: int = 128, kernel_size: int = 14, embed_dim: int = 768,
                 *, convert_to_16: bool = False) -> None:
        super().__init__()
        if convert_to_16:
            msg = "gdlhip DOFAv2Embedding: convert_to_16 (bicubic kernel resize) is not on the hot path"
            raise NotImplementedError(msg)
        self.dynamic_embed_dim = dynamic_embed_dim
        self.kernel_size = kernel_size
        self.embed_dim = embed_dim
        self.convert_to_16 = convert_to_16
        self._num_kernel = kernel_size * kernel_size * embed_dim
        self.weight_generator = TransformerWeightGenerator(dynamic_embed_dim, self._num_kernel, embed_dim)
        self.fclayer = FCResLayer(dynamic_embed_dim)
        self.scaler = 0.01
        for m in self.modules():  # dofa_v2.py:140-146
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.data.fill_(0.01)

    @staticmethod
    def k_padded(in_channels: int, kernel_size: int, cd: torch.dtype) -> int:
        bke = 32 if cd == torch.float32 else 64
        return (in_channels * kernel_size * kernel_size + bke - 1) // bke * bke

    def generate(self, wavelengths: Tensor) -> tuple[Tensor, Tensor]:
        """Raw generator outputs: weight [C, k*k*D], bias [D] (before the 0.01 scaler)."""
        dev = self.fclayer.w1.weight.device
        # `wavelengths * 1000` on the host copy: same f32 rounding as the reference (dofa_v2.py:152)
        pos = (wavelengths.detach().to("cpu", torch.float32) * 1000).to(dev)
        waves = self.fclayer(position_embedding(self.dynamic_embed_dim, pos))
        return self.weight_generator(waves)

    def dynamic_gemm_operands(self, wavelengths: Tensor, cd: torch.dtype) -> tuple[Tensor, Tensor]:
        """GEMM weight [D, Kpad] (compute dtype) and bias [D] (f32), both times ``scaler``."""
        c = wavelengths.numel()
        weight, bias = self.generate(wavelengths)
        kk = self.kernel_size * self.kernel_size
        w = ops.dofa_pack_kernel(weight, c, kk, self.embed_dim, self.scaler,
                                 self.k_padded(c, self.kernel_size, cd), cd)
        return w, ops.scale_f32(bias, self.scaler)

    def forward(self, x: Tensor, wavelengths: Tensor) -> Tensor:
        """[B,C,H,W] -> [B, L, D] f32 (dofa_v2.py:148-181)."""
        cd = gnn.compute_dtype()
        b, c, h, w_ = x.shape
        k = self.kernel_size
        gh, gw = (h + 2 - k) // k + 1, (w_ + 2 - k) // k + 1
        wq, bias = self.dynamic_gemm_operands(wavelengths, cd)
        cols = ops.patchify(ops.image_f32(x, "DOFAv2"), k, 1, gh, gw, wq.shape[1], cd)
        return ops.linear(cols, wq, bias, out_dtype=torch.float32).view(b, gh * gw, self.embed_dim)


class _Attention(nn.Module):
    """timm Attention parameter container (qkv_bias=True)."""

    def __init__(self, dim: int, num_heads: int) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float) -> None:
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    """timm 1.0.24 ``vision_transformer.Block`` as instantiated at dofa_v2.py:250-260
    (LayerNorm eps 1e-5, LayerScale, per-sample DropPath); SURVEY.md Appendix A.1."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, drop_path: float = 0.0,
                 norm_layer=nn.LayerNorm, init_values: float | None = 1e-5) -> None:
        super().__init__()
        if not qkv_bias or proj_drop or attn_drop or init_values is None:
            msg = "gdlhip Block implements the DOFA configuration (qkv_bias, LayerScale, no dropout)"
            raise NotImplementedError(msg)
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads)
        self.ls1 = _LayerScale(dim, init_values)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = _LayerScale(dim, init_values)
        self.drop_prob = float(drop_path)

    def _drop_scale(self, batch: int, device, mask: Tensor | None) -> Tensor | None:
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        if mask is None:
            mask = torch.empty(batch, device=device, dtype=torch.float32).bernoulli_(keep)
        return (mask.to(device=device, dtype=torch.float32) / keep).contiguous()

    def forward(self, x: Tensor, masks: tuple[Tensor, Tensor] | None = None,
                scales: tuple[Tensor | None, Tensor | None] | None = None) -> Tensor:
        """x: f32 [B, N, D] token stream.  ``masks`` pins the two DropPath draws (tests); ``scales`` = the two per-sample
        scales already drawn for this block (gdlhip.nn.drop_path_scales: one launch for the whole encoder).
        One autograd node per block (gdlhip.tnn._VitBlock): forward and backward are HIP kernels."""
        b = x.shape[0]
        if scales is not None and masks is None:
            s1, s2 = scales
        else:
            s1 = self._drop_scale(b, x.device, None if masks is None else masks[0])
            s2 = self._drop_scale(b, x.device, None if masks is None else masks[1])
        at, m = self.attn, self.mlp
        prm = (self.norm1.weight, self.norm1.bias, at.qkv.weight, at.qkv.bias, at.proj.weight, at.proj.bias,
               self.ls1.gamma, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
               self.ls2.gamma)
        return tnn.vit_block(x, s1, s2, at.num_heads, self.norm1.eps, gnn.compute_dtype(), prm)


class DOFAv2(nn.Module):
    """Dynamic One-For-All v2 encoder (dofa_v2.py:184-501)."""

    def __init__(  # noqa: PLR0913
        self,
        encoder_name: str = "dofa_base",
        img_size: int | tuple[int, int] = 224,
        patch_size: int = 14,
        embed_dim: int = 768,
        depth: int = 12,
        num_heads: int = 12,
        mlp_ratio: float = 4.0,
        drop_rate: float = 0.0,
        drop_path_rate: float = 0.1,
        out_indices: list[int] | None = None,
        norm_layer: nn.Module = nn.LayerNorm,
        init_values: float = 1e-5,
        *,
        convert_patch_to_16: bool = False,
        pretrained: bool = True,
    ) -> None:
        super().__init__()
        if drop_rate != 0.0:
            msg = "gdlhip DOFAv2: drop_rate must be 0 (the reference's value)"
            raise NotImplementedError(msg)
        self.encoder_name = encoder_name
        self.pretrained = pretrained
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.img_size = tuple(img_size)
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.depth = depth
        self.num_heads = num_heads
        self.mlp_ratio = mlp_ratio
        eff = 16 if convert_patch_to_16 else patch_size
        self.num_patches = (img_size[0] // eff) * (img_size[1] // eff)
        self.out_indices = out_indices if out_indices is not None else [depth - 1]
        self.patch_embed = DOFAv2Embedding(128, patch_size, embed_dim, convert_to_16=convert_patch_to_16)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim), requires_grad=False)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=True,
                  proj_drop=drop_rate, attn_drop=drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                  init_values=init_values) for i in range(depth)])
        self.norm = norm_layer(embed_dim)  # kept for state-dict parity; unused in forward (:478-486)
        self.init_weights()
        if self.pretrained:
            self.load_pretrained_weights()
        self._dyn_cache: dict = {}

    def init_weights(self) -> None:
        """2-D sincos pos_embed + N(0, 0.02) cls token (dofa_v2.py:268-284): host-side init."""
        pos = self.get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.num_patches**0.5),
                                           cls_token=True)
        self.pos_embed.data.copy_(pos.unsqueeze(0))
        nn.init.normal_(self.cls_token, std=0.02)

    CHECKPOINT_FILES = {"dofa_base": "dofav2_vit_base_e150.pth", "dofa_large": "dofav2_vit_large_e150.pth"}

    def load_pretrained_weights(self):
        """dofa_v2.py:286-347.  The reference fetches ``hf.co/earthflow/DOFA/.../dofav2_vit_{base,large}_e150.pth`` with
        ``torch.hub.load_state_dict_from_url``, which caches the file under ``<torch hub dir>/checkpoints``.  This build
        never opens a network connection: it reads that same cache location (or ``$GDL_DOFA_CHECKPOINT``) and fails
        loudly, naming the file it wants, when the checkpoint is not there."""
        import os
        from pathlib import Path
        if self.encoder_name not in self.CHECKPOINT_FILES:
            msg = f"Unknown model name: {self.encoder_name}"
            raise ValueError(msg)
        name = self.CHECKPOINT_FILES[self.encoder_name]
        candidates = [os.environ.get("GDL_DOFA_CHECKPOINT"), str(Path(torch.hub.get_dir()) / "checkpoints" / name)]
        for path in candidates:
            if path and Path(path).is_file():
                return self.load_pretrained_state_dict(torch.load(path, map_location="cpu", weights_only=True))
        msg = (f"pretrained=True needs the published checkpoint {name} (https://hf.co/earthflow/DOFA); this build does not "
               f"download: put it at {candidates[1]} or point GDL_DOFA_CHECKPOINT at it, or pass pretrained=False")
        raise RuntimeError(msg)

    def load_pretrained_state_dict(self, state_dict: dict) -> tuple[list[str], list[str]]:
        """Key remapping of the published DOFA checkpoints (dofa_v2.py:303-347)."""
        if "model" in state_dict:
            state_dict = state_dict["model"]
        new_sd = {}
        for key, value in state_dict.items():
            new_key = key
            if key.startswith("model."):
                new_key = key[6:]
                if not (new_key.startswith(("blocks.", "norm.")) or new_key in {"cls_token", "pos_embed"}):
                    continue
            new_sd[new_key] = value
        if "pos_embed" in new_sd and self.pos_embed.shape != new_sd["pos_embed"].shape:
            new_sd["pos_embed"] = self._resize_pos_embed(new_sd["pos_embed"], self.num_patches,
                                                         self.num_patches + 1)
        missing, unexpected = self.load_state_dict(new_sd, strict=False)
        actual_missing = set(missing) - {"head.weight", "head.bias"}
        if actual_missing:
            msg = f"Missing required keys in state dict: {actual_missing}"
            raise RuntimeError(msg)
        if unexpected:
            msg = f"Unexpected keys in state dict: {unexpected}"
            raise RuntimeError(msg)
        return missing, unexpected

    def _resize_pos_embed(self, pos_embed: Tensor, num_patches: int, num_tokens: int) -> Tensor:
        """Load-time bicubic resize of the position grid (dofa_v2.py:349-392); host-side, once."""
        if pos_embed.shape[1] == num_tokens:
            return pos_embed
        cls_token, pos_tokens = pos_embed[:, :1, :], pos_embed[:, 1:, :]
        old, new = int(pos_tokens.shape[1] ** 0.5), int(num_patches**0.5)
        if old != new:
            grid = pos_tokens.reshape(1, old, old, -1).permute(0, 3, 1, 2)
            grid = torch.nn.functional.interpolate(grid, size=(new, new), mode="bicubic",
                                                   align_corners=False)
            pos_tokens = grid.permute(0, 2, 3, 1).reshape(1, -1, pos_embed.shape[-1])
        return torch.cat([cls_token, pos_tokens], dim=1)

    @staticmethod
    def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int, *, cls_token: bool = False) -> Tensor:
        grid = torch.meshgrid(torch.arange(grid_size), torch.arange(grid_size), indexing="ij")
        grid = torch.stack(grid, dim=0).reshape([2, 1, grid_size, grid_size])
        pos_embed = DOFAv2.get_2d_sincos_pos_embed_from_grid(embed_dim, grid)
        if cls_token:
            pos_embed = torch.cat([torch.zeros([1, embed_dim]), pos_embed], dim=0)
        return pos_embed

    @staticmethod
    def get_2d_sincos_pos_embed_from_grid(embed_dim: int, grid: Tensor) -> Tensor:
        if embed_dim % 2 != 0:
            msg = "embed_dim must be even"
            raise ValueError(msg)
        emb_h = DOFAv2.get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
        emb_w = DOFAv2.get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
        return torch.cat([emb_h, emb_w], dim=1)

    @staticmethod
    def get_1d_sincos_pos_embed_from_grid(embed_dim: int, pos: Tensor) -> Tensor:
        if embed_dim % 2 != 0:
            msg = "embed_dim must be even"
            raise ValueError(msg)
        omega = torch.arange(embed_dim // 2, dtype=torch.float32)
        omega /= embed_dim / 2.0
        omega = 1.0 / 10000**omega
        out = torch.einsum("m,d->md", pos.reshape(-1), omega)
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)

    # ------------------------------------------------------------------ forward
    def _check_wavelengths(self, wavelengths: Tensor) -> tuple[Tensor, tuple]:
        """dofa_v2.py:437-442; the equality check runs on the host copy (no extra device sync
        when the batch dict keeps wavelengths on the host)."""
        host = wavelengths.detach().to("cpu", torch.float32)
        if host.dim() == 2:
            if not torch.allclose(host, host[0:1].expand_as(host)):
                msg = "DOFA cannot handle different wavelengths within a batch"
                raise ValueError(msg)
            host = host[0]
        return host, tuple(host.tolist())

    def _dynamic_operands(self, wavelengths: Tensor, device, cd: torch.dtype):
        host, key = self._check_wavelengths(wavelengths)
        params = tuple(self.patch_embed.parameters())

        def build():
            with torch.no_grad():
                return self.patch_embed.dynamic_gemm_operands(host, cd)

        # depends only on the wavelengths + the generator weights: regenerated when those change
        return gnn.cached(params, f"dofa_dyn:{key}:{cd}", build)

    def _trainable_operands(self, wavelengths: Tensor, cd: torch.dtype) -> tuple[Tensor, Tensor]:
        """(f32 [D, Kpad] patch-embed weight, f32 [D] bias) with gradients into every generator parameter."""
        host, _ = self._check_wavelengths(wavelengths)
        pe = self.patch_embed
        gen = pe.weight_generator
        if len(gen.transformer_encoder.layers) != 1:
            msg = "gdlhip DOFAv2: the generator backward is built for num_layers=1 (the reference's value)"
            raise NotImplementedError(msg)
        layer = gen.transformer_encoder.layers[0]
        sa = layer.self_attn
        dev = pe.fclayer.w1.weight.device
        pos = (host * 1000).to(dev)
        emb = position_embedding(pe.dynamic_embed_dim, pos)
        prm = (pe.fclayer.w1.weight, pe.fclayer.w1.bias, pe.fclayer.w2.weight, pe.fclayer.w2.bias, gen.weight_tokens,
               gen.bias_token, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
               layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm1.weight,
               layer.norm1.bias, layer.norm2.weight, layer.norm2.bias, gen.fc_weight.weight, gen.fc_weight.bias,
               gen.fc_bias.weight, gen.fc_bias.bias)
        kpad = pe.k_padded(host.numel(), pe.kernel_size, cd)
        return tnn.dofa_generator(emb, sa.num_heads, pe.kernel_size * pe.kernel_size, pe.embed_dim, pe.scaler, kpad,
                                  layer.norm1.eps, layer.norm2.eps, prm)

    def _tokens(self, x: Tensor, wavelengths: Tensor, drop_masks):
        """Generator of (block index, f32 token stream [B, 1+n, D]) after every block."""
        cd = gnn.compute_dtype()
        b, c, h, w_ = x.shape
        k = self.patch_size
        gh, gw = (h + 2 - k) // k + 1, (w_ + 2 - k) // k + 1
        n = gh * gw
        if n + 1 != self.pos_embed.shape[1]:
            msg = f"image {h}x{w_} gives {n} patches but pos_embed has {self.pos_embed.shape[1] - 1}"
            raise ValueError(msg)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.patch_embed.parameters()):
            wq, bias = self._trainable_operands(wavelengths, cd)       # one autograd node (gdlhip.tnn._DofaGenerator)
        else:
            wq, bias = self._dynamic_operands(wavelengths, x.device, cd)
        cols = ops.patchify(ops.image_f32(x, "DOFAv2"), k, 1, gh, gw, wq.shape[1], cd)
        # cls token rows (no pos-embed: dofa_v2.py:447-452), then patch GEMM + bias + pos_embed[1:]
        tok = tnn.dofa_tokens(cols.view(b, n, -1), wq, bias, self.cls_token, self.pos_embed)
        scales = None
        if self.training and drop_masks is None:      # every DropPath draw of the pass in one launch
            scales = gnn.drop_path_scales([blk.drop_prob for blk in self.blocks], b, tok.device)
        for i, blk in enumerate(self.blocks):
            tok = blk(tok, None if drop_masks is None else drop_masks[i], None if scales is None else scales[i])
            yield i, tok

    def forward_features_nhwc(self, x: Tensor, wavelengths: Tensor, drop_masks=None) -> list[Tensor]:
        """Feature taps as NHWC tensors in the compute dtype (what the neck kernels consume)."""
        cd = gnn.compute_dtype()
        feats = []
        for i, tok in self._tokens(x, wavelengths, drop_masks):
            if i in self.out_indices:
                feats.append(tnn.tap(tok, int((tok.shape[1] - 1) ** 0.5), cd))
        return feats

    def forward_features(self, x: Tensor, wavelengths: Tensor,
                         drop_masks: list[tuple[Tensor, Tensor]] | None = None) -> list[Tensor]:
        """Reference API: f32 NCHW taps, zero-copy channels-last views of the token stream."""
        feats = []
        for i, tok in self._tokens(x, wavelengths, drop_masks):
            if i in self.out_indices:
                hw = int((tok.shape[1] - 1) ** 0.5)
                feats.append(tok[:, 1:, :].unflatten(1, (hw, hw)).permute(0, 3, 1, 2))
        return feats

    def forward(self, x: Tensor, wavelengths: Tensor, drop_masks=None) -> list[Tensor]:
        return self.forward_features(x, wavelengths, drop_masks)


def create_dofa_base(img_size=224, out_indices=None, *, pretrained: bool = True, **kwargs) -> DOFAv2:
    """dofa_v2.py:504-534."""
    return DOFAv2(encoder_name="dofa_base", img_size=img_size, patch_size=14, embed_dim=768,
                  num_heads=12, depth=12, out_indices=out_indices or [4, 6, 10, 11],
                  pretrained=pretrained, **kwargs)


def create_dofa_large(img_size=224, out_indices=None, *, pretrained: bool = True, **kwargs) -> DOFAv2:
    """dofa_v2.py:537-567."""
    return DOFAv2(encoder_name="dofa_large", img_size=img_size, patch_size=14, embed_dim=1024,
                  num_heads=16, depth=24, out_indices=out_indices or [5, 9, 15, 21],
                  pretrained=pretrained, **kwargs)
