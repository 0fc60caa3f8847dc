"""Trainable transformer building blocks: conv / linear, LayerNorm, and whole MiT / ViT blocks as
single autograd nodes whose forward AND backward are sequences of libgdlhip.so kernels.

A block-level ``torch.autograd.Function`` (instead of one node per op) lets the backward fuse the
residual-stream gradient adds into the LayerNorm-backward kernel and the two data-gradient
contributions of the attention input into one GEMM epilogue, and keeps autograd bookkeeping to one
node per transformer block.

Reference semantics: MiT Block mix_transformer.py:160-221 (Attention :66-157, Mix-FFN :17-63,
DWConv :533-546, OverlapPatchEmbed :224-276); timm ViT Block as used by dofa_v2.py:248-263.
"""

from __future__ import annotations

import torch
from torch import Tensor
from torch.autograd import Function

from . import ops
from .nn import cached, conv_weight_matrix, gemm_weight, to_compute
from .ops import ACT_GELU, ACT_MUL_GELU_GRAD, ACT_NONE


# ------------------------------------------------------------------ operand helpers
def _wmat32(weight: Tensor) -> Tensor:
    """f32 [N, K] GEMM view of a Linear ([N,K]) or conv ([N,C,R,S] -> K = (r,s,c)) parameter."""
    w = weight.detach() if weight.dim() == 2 else conv_weight_matrix(weight)
    return w if w.is_contiguous() else w.contiguous()


def t_weight(weight: Tensor, cd: torch.dtype) -> Tensor:
    """[K, N] transposed operand (compute dtype) for the data gradient of a Linear / strided conv:
    dx_cols[m, k] = sum_n dy[m, n] * w[n, k]."""
    def build():
        wm = _wmat32(weight)
        return ops.pack_dgrad(wm, wm.shape[0], 1, wm.shape[1], cd)
    return cached((weight,), f"wT:{cd}", build)


def flipped_weight(weight: Tensor, cd: torch.dtype) -> Tensor:
    """[C, (R*S flipped)*N] operand for the data gradient of a stride-1 conv."""
    def build():
        n, c, r, s = weight.shape
        return ops.pack_dgrad(conv_weight_matrix(weight), n, r * s, c, cd)
    return cached((weight,), f"dgrad:{cd}", build)


def _dense(g: Tensor) -> Tensor:
    """Contiguous version of a (possibly strided, unit channel stride) gradient (ops.copy_cast when needed)."""
    if g.is_contiguous():
        return g
    shape = g.shape
    g4 = g if g.dim() == 4 else g.reshape(shape[0], 1, -1, shape[-1]) if g.dim() == 3 else g.unsqueeze(0).unsqueeze(0)
    return ops.copy_cast(g4, out_dtype=g.dtype).view(shape)


def _to_cd(g: Tensor, cd: torch.dtype) -> Tensor:
    """Dense tensor in the compute dtype (one kernel: cast, or strided copy+cast)."""
    if g.dtype == cd:
        return _dense(g)
    if g.is_contiguous():
        return ops.cast(g, cd)
    shape = g.shape
    g4 = g if g.dim() == 4 else g.reshape(shape[0], 1, -1, shape[-1])
    return ops.copy_cast(g4, out_dtype=cd).view(shape)


def _conv_shape(weight: Tensor) -> tuple[int, int, int, int]:
    if weight.dim() == 2:
        return weight.shape[0], weight.shape[1], 1, 1
    return tuple(weight.shape)  # n, c, r, s


def _as_param_grad(dw: Tensor, weight: Tensor) -> Tensor:
    """f32 [N, (r,s,c)] weight gradient -> a tensor of the parameter's logical shape."""
    if weight.dim() == 2:
        return dw
    n, c, r, s = weight.shape
    if r == 1 and s == 1:
        return dw.view(n, c, 1, 1)          # the strides torch keeps for a channels-last 1x1 parameter
    return dw.view(n, r, s, c).permute(0, 3, 1, 2)


# ------------------------------------------------------------------ raw backward pieces (no autograd)
def linear_dx(dz: Tensor, weight: Tensor, *, resid: Tensor | None = None, act: int = ACT_NONE,
              out_dtype: torch.dtype | None = None) -> Tensor:
    """dz [..., N] (compute dtype) -> dz @ W [..., K]; ``resid`` is added (or, with
    ACT_MUL_GELU_GRAD, multiplied as gelu'(resid)) in the GEMM epilogue."""
    return ops.linear(dz, t_weight(weight, dz.dtype), None, resid=resid, act=act, out_dtype=out_dtype)


def linear_dw(x: Tensor, dz: Tensor) -> tuple[Tensor, Tensor]:
    """(dW [N,K] f32, db [N] f32) of y = x W^T + b from x [..., K], dz [..., N]."""
    x2, d2 = x.reshape(-1, x.shape[-1]), dz.reshape(-1, dz.shape[-1])
    return ops.conv_wgrad(x2, d2, R=1, S=1), ops.colsum(d2)


def conv_dx(dy: Tensor, weight: Tensor, stride: int, pad: int, in_hw: tuple[int, int]) -> Tensor:
    """Data gradient of conv(x NHWC, weight, stride, pad) given dense dy NHWC (compute dtype)."""
    n, c, r, s = _conv_shape(weight)
    if stride == 1:
        if r == 1 and s == 1:
            return linear_dx(dy, weight)
        return ops.conv_gemm(dy, flipped_weight(weight, dy.dtype), R=r, S=s, pad=r - 1 - pad)
    b, ho, wo, _ = dy.shape
    cols = ops.linear(dy.reshape(-1, n), t_weight(weight, dy.dtype), None)       # [B*Ho*Wo, (r,s,c)]
    return ops.col2im(cols, b, ho, wo, r, s, c, stride, pad, in_hw[0], in_hw[1], dy.dtype)


# ------------------------------------------------------------------ conv / linear node
class _Conv(Function):
    """y = conv(x, weight) + bias on NHWC (Linear = 1x1 on [B,1,N,K]); output dtype selectable."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_dtype):
        n, c, r, s = _conv_shape(weight)
        y = ops.conv_gemm(x, gemm_weight(weight, x.dtype), R=r, S=s, stride=stride, pad=pad,
                          bias=None if bias is None else bias.detach(), out_dtype=out_dtype)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        n, c, r, s = _conv_shape(weight)
        dy = _to_cd(g, x.dtype)
        dw = db = dx = None
        if ctx.needs_input_grad[1]:
            dw = _as_param_grad(ops.conv_wgrad(x, dy, R=r, S=s, stride=stride, pad=pad), weight)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy)
        if ctx.needs_input_grad[0]:
            dx = conv_dx(dy, weight, stride, pad, (x.shape[1], x.shape[2]))
        return dx, dw, db, None, None, None


def conv(x: Tensor, weight: Tensor, bias: Tensor | None, *, stride: int = 1, pad: int = 0,
         out_dtype: torch.dtype | None = None) -> Tensor:
    """Differentiable conv / 1x1 on an NHWC tensor in the compute dtype."""
    return _Conv.apply(x, weight, bias, stride, pad, out_dtype)


def linear(x: Tensor, weight: Tensor, bias: Tensor | None, out_dtype: torch.dtype | None = None) -> Tensor:
    """Differentiable nn.Linear on [..., K] (compute dtype)."""
    lead = x.shape[:-1]
    x4 = x.reshape(1, 1, -1, x.shape[-1]) if x.dim() != 4 else x
    y = _Conv.apply(x4, weight, bias, 1, 0, out_dtype)
    return y.reshape(*lead, weight.shape[0])


class _StemLinear(Function):
    """Patchified image stem: y = cols @ pad(W.reshape(N, c*k*k))^T + b (raw bands carry no gradient)."""

    @staticmethod
    def forward(ctx, cols, weight, bias, wq, out_dtype):
        y = ops.linear(cols, wq, bias.detach(), out_dtype=out_dtype)
        ctx.save_for_backward(cols, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        cols, weight = ctx.saved_tensors
        dy = _to_cd(g, cols.dtype)
        dwp, db = linear_dw(cols, dy)                                  # [N, kpad]
        n = weight.shape[0]
        k = weight[0].numel()
        dw = dwp[:, :k].reshape(weight.shape)                          # (c, r, s) order = logical OIHW
        return None, dw, db, None, None


def stem_linear(cols: Tensor, weight: Tensor, bias: Tensor, wq: Tensor, out_dtype: torch.dtype) -> Tensor:
    return _StemLinear.apply(cols, weight, bias, wq, out_dtype)


class _StemConv(Function):
    """Image stem WITHOUT an im2col matrix: y = conv_kxk/stride(image) + b evaluated as sub-pixel-phase 3x3 convolutions on the
    4 x 4 space-to-depth image (gdlhip.cnn.mark_stem; MiT's 7x7 / 4 is a single phase).  Raw bands carry no gradient."""

    @staticmethod
    def forward(ctx, xs, weight, bias, out_dtype):
        from . import cnn
        n = weight.shape[0]
        wq = cnn.stem_operands(weight, xs.dtype, xs.shape[-1], n)
        st, _ = weight._gdl_stem
        e = cnn.STEM_BLOCK // st
        b, hb, wb, _ = xs.shape
        y = torch.empty((b, hb * e, wb * e, n), device=xs.device, dtype=out_dtype)
        for ph in range(e * e):
            ops.conv_gemm(xs, wq[ph], R=3, S=3, pad=1, bias=bias.detach(), out=y[:, ph // e::e, ph % e::e, :])
        ctx.save_for_backward(xs, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import cnn
        xs, weight = ctx.saved_tensors
        dy = _dense(_to_cd(g, xs.dtype))
        db = ops.colsum(dy.reshape(-1, dy.shape[-1]))
        return None, cnn.stem_param_grad(xs, dy, weight), db, None


def stem_conv(xs: Tensor, weight: Tensor, bias: Tensor, out_dtype: torch.dtype) -> Tensor:
    """xs = gdlhip.cnn.space_to_depth_image(image); weight marked with gdlhip.cnn.mark_stem(weight, stride, pad)."""
    return _StemConv.apply(xs, weight, bias, out_dtype)


# ------------------------------------------------------------------ dynamic SegFormer stem: band weighting + pooling
class _ChanPool(Function):
    """agg = sum_c softmax_c(s)[c] * conv[c] * cw[c] (DynamicChannelEmbed.forward, mix_transformer.py:823-853) with the
    band weights cw and the attention biases generated from the position codes (:781-801) inside the node."""

    @staticmethod
    def forward(ctx, conv, pos, wg0w, wg0b, wg2w, wg2b, ca0w, ca0b, ca2w, ca2b):
        W1 = ca0w.detach().reshape(ca0w.shape[0], -1)            # Conv1d(k=1) weight [H1, E + PD, 1]
        w2 = ca2w.detach().reshape(-1)
        hid, cw, hb = ops.chan_weights(pos, wg0w.detach(), wg0b.detach(), wg2w.detach(), wg2b.detach(), W1, ca0b.detach())
        b2s = 0.0   # channel_attention[2].bias shifts every band's logit alike: the softmax (and so agg) ignores it
        agg, _ = ops.chan_pool(conv, cw, W1, hb, w2, b2s)
        ctx.save_for_backward(conv, pos, wg2w, hid, cw, hb, W1, w2)
        ctx.b2s = b2s
        ctx.shapes = (ca0w.shape, ca2w.shape, ca2b.shape)
        return agg

    @staticmethod
    def backward(ctx, g):
        conv, pos, wg2w, hid, cw, hb, W1, w2 = ctx.saved_tensors
        dconv, dW1a, dhb, dw2, dcw = ops.chan_pool_bwd(conv, cw, W1, hb, w2, ctx.b2s, _dense(g).float())
        dW0, db0, dW2, db2, dW1b, db1 = ops.chan_weights_bwd(pos, wg2w.detach(), hid, cw, dcw.contiguous(), dhb.contiguous())
        s0, s2, sb = ctx.shapes
        dca0 = torch.cat([dW1a, dW1b], dim=1).reshape(s0)
        return (dconv, None, dW0, db0, dW2, db2, dca0, db1, dw2.reshape(s2),
                torch.zeros(sb, device=g.device, dtype=torch.float32))


def chan_pool(conv: Tensor, pos: Tensor, weight_gen, channel_attention) -> Tensor:
    """conv [B, C, P, E] f32 (shared spatial conv of every band) -> pooled tokens [B, P, E] f32."""
    return _ChanPool.apply(conv, pos, weight_gen[0].weight, weight_gen[0].bias, weight_gen[2].weight, weight_gen[2].bias,
                           channel_attention[0].weight, channel_attention[0].bias, channel_attention[2].weight,
                           channel_attention[2].bias)


# ------------------------------------------------------------------ LayerNorm node
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        return ops.layernorm(x, weight.detach(), bias.detach(), eps, out_dtype)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, _dense(g), weight.detach(), ctx.eps)
        return dx, dg, db, None, None


def layernorm(x: Tensor, norm: torch.nn.LayerNorm, out_dtype: torch.dtype) -> Tensor:
    """Differentiable F.layer_norm on the f32 token stream."""
    return _LayerNorm.apply(x, norm.weight, norm.bias, norm.eps, out_dtype)


# ------------------------------------------------------------------ MiT block
class _MitBlock(Function):
    """SegFormer block: x + dp(proj(SRA(LN1 x))) then + dp(fc2(gelu(dw3x3(fc1(LN2 .))))).

    inputs: x f32 [B,N,C]; s1, s2 DropPath batch scales ([B] f32 or None); then the parameters
    n1w n1b qw qb kvw kvb pw pb n2w n2b f1w f1b dww dwb f2w f2b [srw srb nsw nsb].
    """

    @staticmethod
    def forward(ctx, x, s1, s2, hh, ww, heads, sr, eps1, eps_sr, cd, *prm):
        n1w, n1b, qw, qb, kvw, kvb, pw, pb, n2w, n2b, f1w, f1b, dww, dwb, f2w, f2b = prm[:16]
        b, n, c = x.shape
        d = [p.detach() for p in prm]
        h1 = ops.layernorm(x, d[0], d[1], eps1, cd)
        q = ops.linear(h1, gemm_weight(qw, cd), d[3])
        red = None
        if sr > 1:
            srw, srb, nsw, nsb = prm[16:20]
            red = ops.conv_gemm(h1.view(b, hh, ww, c), gemm_weight(srw, cd), R=sr, S=sr, stride=sr, bias=d[17],
                                out_dtype=torch.float32)
            xn = ops.layernorm(red.view(b, -1, c), d[18], d[19], eps_sr, cd)
        else:
            xn = h1
        kv = ops.linear(xn, gemm_weight(kvw, cd), d[5])
        a, lse = ops.attention(q, kv[..., :c], kv[..., c:], heads, return_lse=True)
        x1 = torch.empty_like(x)
        ops.conv_gemm(a.view(b, 1, n, c), gemm_weight(pw, cd), bias=d[7], batch_scale=s1, resid=x.view(b, 1, n, c),
                      out=x1.view(b, 1, n, c))
        h2 = ops.layernorm(x1, d[8], d[9], eps1, cd)
        u = ops.linear(h2, gemm_weight(f1w, cd), d[11])
        w9 = cached((dww,), "dw9", lambda: dww.detach().reshape(dww.shape[0], 9).t().contiguous())
        g = ops.dwconv3x3(u.view(b, hh, ww, -1), w9, d[13], True)
        x2 = torch.empty_like(x)
        ops.conv_gemm(g.view(b, 1, n, -1), gemm_weight(f2w, cd), bias=d[15], batch_scale=s2,
                      resid=x1.view(b, 1, n, c), out=x2.view(b, 1, n, c))
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, s1, s2, h1, q, red, xn if sr > 1 else None, kv, a, x1, h2, u, g, w9, lse, *prm)
            ctx.cfg = (hh, ww, heads, sr, eps1, eps_sr, cd)
        return x2

    @staticmethod
    def backward(ctx, gx2):
        x, s1, s2, h1, q, red, xn, kv, a, x1, h2, u, g, w9, lse, *prm = ctx.saved_tensors
        hh, ww, heads, sr, eps1, eps_sr, cd = ctx.cfg
        n1w, n1b, qw, qb, kvw, kvb, pw, pb, n2w, n2b, f1w, f1b, dww, dwb, f2w, f2b = prm[:16]
        b, n, c = x.shape
        gx2 = _dense(gx2)
        # ---- Mix-FFN branch
        dz2, _ = ops.layerscale_bwd(gx2, None, None, s2, cd)
        df2w, df2b = linear_dw(g, dz2)
        dg = linear_dx(dz2, f2w)
        du, dw9, ddwb = ops.dwconv3x3_gelu_bwd(u.view(b, hh, ww, -1), dg.view(b, hh, ww, -1), w9, dwb.detach())
        du = du.view(b, n, -1)
        df1w, df1b = linear_dw(h2, du)
        dh2 = linear_dx(du, f1w)
        gx1, dn2w, dn2b = ops.layernorm_bwd(x1, dh2, n2w.detach(), eps1, dres=gx2)
        # ---- attention branch
        dz1, _ = ops.layerscale_bwd(gx1, None, None, s1, cd)
        dpw, dpb = linear_dw(a, dz1)
        da = linear_dx(dz1, pw)
        dq = torch.empty((b, n, c), device=x.device, dtype=cd)
        dkv = torch.empty_like(kv)
        ops.attention_bwd(q, kv[..., :c], kv[..., c:], da, heads, dq, dkv[..., :c], dkv[..., c:], o=a, lse=lse)
        kv_in = xn if sr > 1 else h1
        dkvw, dkvb = linear_dw(kv_in, dkv)
        dxn = linear_dx(dkv, kvw)
        dqw, dqb = linear_dw(h1, dq)
        extra = ()
        if sr > 1:
            srw, srb, nsw, nsb = prm[16:20]
            dred, dnsw, dnsb = ops.layernorm_bwd(red.view(b, -1, c), dxn, nsw.detach(), eps_sr)
            dred = (dred if cd == torch.float32 else ops.cast(dred, cd)).view(red.shape)
            h1_img = h1.view(b, hh, ww, c)
            dsrw = _as_param_grad(ops.conv_wgrad(h1_img, dred, R=sr, S=sr, stride=sr), srw)
            dsrb = ops.colsum(dred)
            side = conv_dx(dred, srw, sr, 0, (hh, ww)).view(b, n, c)
            extra = (dsrw, dsrb, dnsw, dnsb)
        else:
            side = dxn
        dh1 = linear_dx(dq, qw, resid=side)
        gx, dn1w, dn1b = ops.layernorm_bwd(x, dh1, n1w.detach(), eps1, dres=gx1)
        ddww = dw9.t().reshape(dww.shape)
        return (gx, None, None, None, None, None, None, None, None, None,
                dn1w, dn1b, dqw, dqb, dkvw, dkvb, dpw, dpb, dn2w, dn2b, df1w, df1b, ddww, ddwb, df2w, df2b, *extra)


def mit_block(x: Tensor, s1, s2, hh: int, ww: int, heads: int, sr: int, eps1: float, eps_sr: float,
              cd: torch.dtype, params: tuple) -> Tensor:
    return _MitBlock.apply(x, s1, s2, hh, ww, heads, sr, eps1, eps_sr, cd, *params)


# ------------------------------------------------------------------ timm ViT block (DOFA)
class _VitBlock(Function):
    """timm Block with LayerScale + DropPath: x + s1*g1*proj(MHA(LN1 x)); + s2*g2*fc2(gelu(fc1(LN2 .))).

    inputs: x f32 [B,N,C]; s1, s2; parameters n1w n1b qkvw qkvb pw pb g1 n2w n2b f1w f1b f2w f2b g2
    (g1 / g2 may be None when the block has no LayerScale).
    """

    @staticmethod
    def forward(ctx, x, s1, s2, heads, eps, cd, *prm):
        n1w, n1b, qkvw, qkvb, pw, pb, g1, n2w, n2b, f1w, f1b, f2w, f2b, g2 = prm
        b, n, c = x.shape
        d = [None if p is None else p.detach() for p in prm]
        h1 = ops.layernorm(x, d[0], d[1], eps, cd)
        qkv = ops.linear(h1, gemm_weight(qkvw, cd), d[3])
        qv, kv_, vv = ops.split_qkv(qkv)
        a, lse = ops.attention(qv, kv_, vv, heads, return_lse=True)
        # the backward needs the branch outputs before LayerScale (z) and the GELU input (u): extra epilogue
        # stores that are skipped when nothing upstream or in the block wants a gradient (inference, frozen)
        train = any(ctx.needs_input_grad)
        x1, z1 = torch.empty_like(x), (torch.empty_like(x) if train else None)
        ops.conv_gemm(a.view(b, 1, n, c), gemm_weight(pw, cd), bias=d[5], scale=d[6], batch_scale=s1,
                      resid=x.view(b, 1, n, c), out=x1.view(b, 1, n, c),
                      aux_out=z1.view(b, 1, n, c) if train else None)
        h2 = ops.layernorm(x1, d[7], d[8], eps, cd)
        u = torch.empty((b, n, f1w.shape[0]), device=x.device, dtype=cd) if train else None
        f = ops.linear(h2, gemm_weight(f1w, cd), d[10], act=ACT_GELU, aux_out=u)
        x2, z2 = torch.empty_like(x), (torch.empty_like(x) if train else None)
        ops.conv_gemm(f.view(b, 1, n, -1), gemm_weight(f2w, cd), bias=d[12], scale=d[13], batch_scale=s2,
                      resid=x1.view(b, 1, n, c), out=x2.view(b, 1, n, c),
                      aux_out=z2.view(b, 1, n, c) if train else None)
        if train:
            ctx.save_for_backward(x, s1, s2, h1, qkv, a, z1, x1, h2, u, f, z2, lse, *prm)
            ctx.cfg = (heads, eps, cd)
        return x2

    @staticmethod
    def backward(ctx, gx2):
        x, s1, s2, h1, qkv, a, z1, x1, h2, u, f, z2, lse, *prm = ctx.saved_tensors
        n1w, n1b, qkvw, qkvb, pw, pb, g1, n2w, n2b, f1w, f1b, f2w, f2b, g2 = prm
        heads, eps, cd = ctx.cfg
        b, n, c = x.shape
        gx2 = _dense(gx2)
        dz2, dg2 = ops.layerscale_bwd(gx2, z2, None if g2 is None else g2.detach(), s2, cd)
        df2w, df2b = linear_dw(f, dz2)
        du = linear_dx(dz2, f2w, resid=u, act=ACT_MUL_GELU_GRAD)
        df1w, df1b = linear_dw(h2, du)
        dh2 = linear_dx(du, f1w)
        gx1, dn2w, dn2b = ops.layernorm_bwd(x1, dh2, n2w.detach(), eps, dres=gx2)
        dz1, dg1 = ops.layerscale_bwd(gx1, z1, None if g1 is None else g1.detach(), s1, cd)
        dpw, dpb = linear_dw(a, dz1)
        da = linear_dx(dz1, pw)
        dqkv = torch.empty_like(qkv)
        qv, kv_, vv = ops.split_qkv(qkv)
        dq_, dk_, dv_ = ops.split_qkv(dqkv)
        ops.attention_bwd(qv, kv_, vv, da, heads, dq_, dk_, dv_, o=a, lse=lse)
        dqkvw, dqkvb = linear_dw(h1, dqkv)
        dh1 = linear_dx(dqkv, qkvw)
        gx, dn1w, dn1b = ops.layernorm_bwd(x, dh1, n1w.detach(), eps, dres=gx1)
        return (gx, None, None, None, None, None,
                dn1w, dn1b, dqkvw, dqkvb, dpw, dpb, dg1, dn2w, dn2b, df1w, df1b, df2w, df2b, dg2)


def vit_block(x: Tensor, s1, s2, heads: int, eps: float, cd: torch.dtype, params: tuple) -> Tensor:
    return _VitBlock.apply(x, s1, s2, heads, eps, cd, *params)


# ------------------------------------------------------------------ DOFA token assembly / feature taps
class _DofaTokens(Function):
    """tok = [cls ; cols @ wq^T + bias + pos_embed[1:]] as f32 [B, 1+n, D] (dofa_v2.py:444-452)."""

    @staticmethod
    def forward(ctx, cols, wq, bias, cls_token, pos_embed):
        b, n, kp = cols.shape
        d = wq.shape[0]
        tok = torch.empty((b, n + 1, d), device=cols.device, dtype=torch.float32)
        ops.add_rows(cls_token.detach().view(1, d), None, tok[:, 0, :], b)
        wq_c = wq.detach() if wq.dtype == cols.dtype else ops.cast(wq.detach().contiguous(), cols.dtype)
        ops.conv_gemm(cols.view(b, 1, n, kp), wq_c, bias=bias.detach(),
                      resid=pos_embed.detach()[0, 1:, :].view(1, 1, n, d), out=tok[:, 1:, :].unsqueeze(1))
        ctx.save_for_backward(cols)
        return tok

    @staticmethod
    def backward(ctx, g):
        (cols,) = ctx.saved_tensors
        g = _dense(g)
        d = g.shape[-1]
        dcls = ops.colsum(g[:, 0, :]).view(1, 1, d) if ctx.needs_input_grad[3] else None
        dwq = dbias = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dwq, dbias = linear_dw(cols, _to_cd(g[:, 1:, :], cols.dtype))
        return None, dwq, dbias, dcls, None


def dofa_tokens(cols: Tensor, wq: Tensor, bias: Tensor, cls_token: Tensor, pos_embed: Tensor) -> Tensor:
    return _DofaTokens.apply(cols, wq, bias, cls_token, pos_embed)


class _Tap(Function):
    """f32 token stream [B, 1+hw*hw, D] -> NHWC feature [B, hw, hw, D] in the compute dtype (cls row dropped)."""

    @staticmethod
    def forward(ctx, tok, hw, cd):
        ctx.shape = tok.shape
        return ops.bilinear(tok[:, 1:, :].unflatten(1, (hw, hw)), (hw, hw), out_dtype=cd)

    @staticmethod
    def backward(ctx, g):
        gx = torch.zeros(ctx.shape, device=g.device, dtype=torch.float32)
        hw = g.shape[1]
        ops.bilinear(g, (hw, hw), out=gx[:, 1:, :].unflatten(1, (hw, hw)))
        return gx, None, None


def tap(tok: Tensor, hw: int, cd: torch.dtype) -> Tensor:
    return _Tap.apply(tok, hw, cd)


# ------------------------------------------------------------------ DOFA dynamic weight generator
class _DofaGenerator(Function):
    """Wavelengths -> patch-embed GEMM operand, as ONE autograd node (dofa_v2.py:38-181: sincos embedding -> FCRes ->
    [weight tokens ; wave tokens ; bias token] -> one post-norm nn.TransformerEncoderLayer (4 heads, gelu, ffn 2048)
    -> fc_weight / fc_bias -> pack * 0.01).  Everything is tiny (132 tokens x 128) and runs in exact f32.

    inputs: pos_embed E [C,128] (no grad), then the 22 parameters
      0 fc.w1.w 1 fc.w1.b 2 fc.w2.w 3 fc.w2.b 4 weight_tokens 5 bias_token 6 in_proj_w 7 in_proj_b 8 out_proj.w
      9 out_proj.b 10 linear1.w 11 linear1.b 12 linear2.w 13 linear2.b 14 norm1.w 15 norm1.b 16 norm2.w 17 norm2.b
      18 fc_weight.w 19 fc_weight.b 20 fc_bias.w 21 fc_bias.b
    outputs: wq f32 [D, Kpad] (k = c*P*P + p, times scaler), bias f32 [D] (times scaler)."""

    @staticmethod
    def forward(ctx, emb, heads, kk, embed_dim, scaler, kpad, eps1, eps2, *prm):
        f32 = torch.float32
        p = [t.detach() for t in prm]
        c, d = emb.shape
        wt = p[4].shape[0]
        s = wt + c + 1

        def w(i):
            return gemm_weight(prm[i], f32)
        y1 = ops.linear(emb, w(0), p[1], act=ops.ACT_RELU)
        r2 = ops.linear(y1, w(2), p[3], act=ops.ACT_RELU)
        waves = torch.empty_like(emb)
        ops.add_rows(r2, emb, waves, c)
        seq = torch.empty((s, d), device=emb.device, dtype=f32)
        ops.add_rows(p[4], None, seq[:wt], wt)
        ops.add_rows(waves, None, seq[wt:wt + c], c)
        ops.add_rows(p[5], None, seq[s - 1:], 1)
        qkv = ops.linear(seq, w(6), p[7])
        att = ops.attention_unfused(*ops.split_qkv(qkv.unsqueeze(0)), heads)[0]
        h = ops.linear(att, w(8), p[9], resid=seq)
        x1 = ops.layernorm(h, p[14], p[15], eps1, f32)
        u = torch.empty((s, prm[10].shape[0]), device=emb.device, dtype=f32)
        f = ops.linear(x1, w(10), p[11], act=ACT_GELU, aux_out=u)
        h2 = ops.linear(f, w(12), p[13], resid=x1)
        out = ops.layernorm(h2, p[16], p[17], eps2, f32)
        tok = torch.empty((c, d), device=emb.device, dtype=f32)
        ops.add_rows(out[wt:wt + c], waves, tok, c)
        weights = ops.linear(tok, w(18), p[19])
        last = out[s - 1:]
        bias = ops.linear(last, w(20), p[21]).reshape(-1)
        wq = ops.dofa_pack_kernel(weights, c, kk, embed_dim, scaler, kpad, f32)
        ctx.save_for_backward(emb, y1, r2, seq, qkv, att, h, x1, u, f, h2, out, tok, *prm)
        ctx.cfg = (heads, kk, embed_dim, scaler, eps1, eps2, c, wt)
        return wq, ops.scale_f32(bias, scaler)

    @staticmethod
    def backward(ctx, dwq, dbias):
        emb, y1, r2, seq, qkv, att, h, x1, u, f, h2, out, tok, *prm = ctx.saved_tensors
        heads, kk, embed_dim, scaler, eps1, eps2, c, wt = ctx.cfg
        f32 = torch.float32
        s, d = seq.shape
        g = [None] * 22
        dweights = ops.dofa_unpack_grad(dwq.float().contiguous(), c, kk, embed_dim, scaler)      # [C, kk*D]
        dbias_raw = ops.scale_f32(dbias.float().contiguous(), scaler).view(1, -1)
        g[18], g[19] = linear_dw(tok, dweights)
        dtok = linear_dx(dweights, prm[18])                                                      # [C, d]
        last = out[s - 1:]
        g[20], g[21] = linear_dw(last, dbias_raw)
        dlast = linear_dx(dbias_raw, prm[20])
        dout = torch.zeros((s, d), device=seq.device, dtype=f32)
        ops.add_rows(dtok, None, dout[wt:wt + c], c)
        ops.add_rows(dlast, None, dout[s - 1:], 1)
        # post-norm layer, second half: out = LN2(h2), h2 = f W2^T + b2 + x1, f = gelu(u), u = x1 W1^T + b1
        dh2, g[16], g[17] = ops.layernorm_bwd(h2, dout, prm[16].detach(), eps2)
        g[12], g[13] = linear_dw(f, dh2)
        du = linear_dx(dh2, prm[12], resid=u, act=ACT_MUL_GELU_GRAD)
        g[10], g[11] = linear_dw(x1, du)
        dx1 = linear_dx(du, prm[10], resid=dh2)
        # first half: x1 = LN1(h), h = att Wo^T + bo + seq
        dh, g[14], g[15] = ops.layernorm_bwd(h, dx1, prm[14].detach(), eps1)
        g[8], g[9] = linear_dw(att, dh)
        datt = linear_dx(dh, prm[8])
        dqkv = torch.empty_like(qkv)
        q3, k3, v3 = ops.split_qkv(qkv.unsqueeze(0))
        dq3, dk3, dv3 = ops.split_qkv(dqkv.unsqueeze(0))
        ops.attention_bwd(q3, k3, v3, datt.unsqueeze(0), heads, dq3, dk3, dv3)
        g[6], g[7] = linear_dw(seq, dqkv)
        dseq = linear_dx(dqkv, prm[6], resid=dh)
        g[4] = dseq[:wt].contiguous()
        g[5] = dseq[s - 1:].contiguous()
        dwaves = torch.empty((c, d), device=seq.device, dtype=f32)
        ops.add_rows(dseq[wt:wt + c], dtok, dwaves, c)                 # tok = out[...] + waves  and  seq rows = waves
        # FCRes: waves = relu(y1 W2^T + b2) + E, y1 = relu(E W1^T + b1)
        dz2 = ops.relu_bwd(r2, dwaves)
        g[2], g[3] = linear_dw(y1, dz2)
        dz1 = ops.relu_bwd(y1, linear_dx(dz2, prm[2]))
        g[0], g[1] = linear_dw(emb, dz1)
        return (None,) * 8 + tuple(g)


def dofa_generator(emb: Tensor, heads: int, kk: int, embed_dim: int, scaler: float, kpad: int, eps1: float,
                   eps2: float, params: tuple) -> tuple[Tensor, Tensor]:
    return _DofaGenerator.apply(emb, heads, kk, embed_dim, scaler, kpad, eps1, eps2, *params)
