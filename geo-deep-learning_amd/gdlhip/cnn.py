"""Convolutional building blocks of the ResNet-encoder / UNet++ path (smp.UnetPlusPlus; reference call site
tasks_with_models/segmentation_unetplus.py:126-131) as autograd nodes over libgdlhip.so kernels.

Channel granularity: activations carry multiples of 8 (bf16) / 4 (f32) channels, i.e. whole 16-byte pieces; the
implicit-GEMM kernels zero-fill the rest of a 128-byte K chunk themselves, so UNet++'s 32- and 16-channel decoder
stages are stored and normalised at their true width.  A conv whose input has more channels than its weight treats
the extra channels as zero-weight, an output count that is not a multiple of the granularity is padded (zero weight
rows, BN gamma = beta = 0 => exactly 0 after BN/ReLU), and every gradient is sliced back to the parameter's shape.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import nn as gnn
from . import ops
from .nn import (_world, cached, mark_updated, conv_weight_matrix, sync_batch_stats, sync_sum_pair, to_compute,
                 update_running_stats)
from .ops import ACT_NONE, ACT_RELU, ACT_RESID_RELU
from .tnn import _dense


def chunk(cd: torch.dtype) -> int:
    """channels per 128-byte K chunk of the implicit-GEMM kernel"""
    return 32 if cd == torch.float32 else 64


def grain(cd: torch.dtype) -> int:
    """channel granularity of activations: 16-byte pieces (the kernels zero-fill the tail of a K chunk)"""
    return 4 if cd == torch.float32 else 8


def pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def mark_flat(weight: Tensor) -> None:
    """The image stem runs as patchify + GEMM: its [N,C,R,S] parameter is used as the [N, (c,r,s)] matrix.
    The mark lives on the Parameter object itself (ids are recycled after garbage collection)."""
    weight._gdl_flat = True


def mark_groups(weight: Tensor, groups: int) -> None:
    """A grouped convolution's parameter ([N, C / groups, R, S], torchvision ResNeXt ``conv2``).  Round 6: it runs as ONE batched
    implicit-GEMM launch over SUPER-GROUPS of at least 32 channels (``supergroups``: resnext101_32x8d's 8-channel groups four at
    a time as a 32 -> 32 block-diagonal filter, its 16-channel groups two at a time, 32 and 64 channels as they are): forward,
    stride-1 data gradient and weight gradient are batched calls over channel slices (ops.conv_gemm_grouped), 4 x / 2 x / 1 x
    the grouped form's multiply-adds instead of the 32 x of the block-diagonal DENSE filter of round 3 -- which stays the path
    for padded channel counts, the f32 parity runs' narrow cases and the stride-2 data gradient (col2im over a dense GEMM)."""
    weight._gdl_groups = int(groups)


def _groups(weight: Tensor) -> int:
    return getattr(weight, "_gdl_groups", 1)


GROUPED_BATCHED = __import__("os").environ.get("GDL_GROUPED_DENSE", "0") != "1"     # A/B switch: 1 = the round-3 dense form


def supergroups(weight: Tensor, cpad: int, npad: int, cd: torch.dtype) -> tuple[int, int, int, int] | None:
    """(Z, per, c, n) when the grouped parameter can run as Z batched problems of `per` groups each (c input / n output
    channels per problem), else None (dense block-diagonal fallback)."""
    g = _groups(weight)
    if g == 1 or not GROUPED_BATCHED or weight.dim() != 4:
        return None
    nt, cg, _, _ = weight.shape
    ng = nt // g
    if cpad != cg * g or npad != nt:              # padded channel counts: the slices would not line up
        return None
    per = 1 if min(cg, ng) >= 32 else max(1, 32 // min(cg, ng))
    if g % per or (per * cg) % grain(cd) or (per * ng) % grain(cd):
        return None
    return g // per, per, per * cg, per * ng


def grouped_operands(weight: Tensor, cd: torch.dtype, sg: tuple[int, int, int, int]) -> tuple[Tensor, Tensor]:
    """(forward operand [Z, n, R*S*c], stride-1 data-gradient operand [Z, c, (R*S flipped)*n]) in the compute dtype: per
    super-group the block-diagonal filter of its `per` groups."""
    def build():
        Z, per, c, n = sg
        nt, cg, r, s = weight.shape
        ng = nt // (Z * per)
        m = conv_weight_matrix(weight).view(Z, per, ng, r * s, cg)                  # [z, group in z, out, tap, in]
        blk = torch.zeros((Z, per, ng, r * s, per, cg), device=weight.device, dtype=torch.float32)
        idx = torch.arange(per, device=weight.device)
        blk[:, idx, :, :, idx, :] = m.permute(1, 0, 2, 3, 4)                        # group i: outputs i*ng.., inputs i*cg..
        blk = blk.view(Z, n, r * s, c)
        fwd = blk.reshape(Z, n, r * s * c)
        dgr = blk.flip(2).permute(0, 3, 2, 1).reshape(Z, c, r * s * n)              # out = sum_taps dy[. - tap] * w[n, tap, c]
        return fwd.to(cd).contiguous(), dgr.to(cd).contiguous()
    return cached((weight,), f"grpw:{cd}:{sg}", build)


def _grouped_param_grad(dw: Tensor, weight: Tensor, sg: tuple[int, int, int, int]) -> Tensor:
    """[Z, n, R*S*c] f32 -> the parameter's [N, C/groups, R, S]: the block diagonal inside every super-group."""
    Z, per, c, n = sg
    nt, cg, r, s = weight.shape
    ng = nt // (Z * per)
    d = dw.view(Z, per, ng, r * s, per, cg)
    idx = torch.arange(per, device=dw.device)
    d = d[:, idx, :, :, idx, :]                                                     # [per, Z, ng, taps, cg]
    d = d.permute(1, 0, 2, 3, 4).reshape(nt, r, s, cg)
    return d.permute(0, 3, 1, 2)


STEM_BLOCK = 4     # pixels per side of a space-to-depth block
# A/B switch (tools): GDL_STEM_IM2COL=1 restores the strided-patchify (im2col matrix) + GEMM stems of rounds 1 and 2
STEM_IM2COL = __import__("os").environ.get("GDL_STEM_IM2COL", "0") == "1"


def mark_stem(weight: Tensor, stride: int, pad: int) -> None:
    """An image stem's parameter ([N, C, k, k] on raw bands, stride s in {2, 4}: torchvision's 7x7 / 2, MiT's 7x7 / 4).  It runs
    IM2COL-FREE as ordinary 3x3 / stride-1 NHWC convolutions on the SPACE-TO-DEPTH image: 4 x 4 pixel blocks become 16 C
    channels (a pure re-layout of the image by gdl_patchify with patch = stride = 4 -- nothing is duplicated, where the
    strided-patchify matrix this path used before held k*k / s*s copies of every pixel), and the e x e = (4 / s)^2 output
    pixels that share a block are e*e sub-pixel PHASES, each its own 3x3 filter over blocks:
        input pixel  s * (e B + eps) - pad + r  =  4 (B + j) + d      =>      r = 4 j + d - s eps + pad,   j in {-1, 0, 1}
        W_phase[n, (jy, jx), (c, dy, dx)] = W[n, c, r(jy, dy, epsy), r(jx, dx, epsx)]  where that tap exists, else 0.
    Phase (epsy, epsx) writes the output pixels [epsy::e, epsx::e]; the weight gradient is nine-tap row-segment
    weight-gradient calls on the same strided views, scattered back to the k x k taps (stem_operands / stem_param_grad)."""
    if STEM_BLOCK % stride:
        raise ValueError(f"stem stride {stride} does not divide the space-to-depth block {STEM_BLOCK}")
    weight._gdl_stem = (int(stride), int(pad))


_STEM_TABLES: dict = {}


def _stem_index(weight: Tensor):
    """(e, idx, valid): idx[phase, (jy, jx), (c, dy, dx)] = flat index into W[n].reshape(C * k * k) of the source tap.
    Built once per (C, k, stride, pad, device)."""
    key = (weight.shape[1], weight.shape[2], *weight._gdl_stem, str(weight.device))
    hit = _STEM_TABLES.get(key)
    if hit is None:
        e, idx, valid = _stem_index_build(weight)
        # per phase: the filter slots that hold a tap and the taps they hold, as index lists (a boolean mask would cost a
        # device -> host synchronisation per use)
        slots = [valid[ph].reshape(-1).nonzero().reshape(-1) for ph in range(e * e)]
        taps = [idx[ph].reshape(-1)[slots[ph]] for ph in range(e * e)]
        hit = _STEM_TABLES[key] = (e, idx, valid, slots, taps)
    return hit[:3]


def _stem_lists(weight: Tensor):
    _stem_index(weight)
    hit = _STEM_TABLES[(weight.shape[1], weight.shape[2], *weight._gdl_stem, str(weight.device))]
    return hit[3], hit[4]


def _stem_index_build(weight: Tensor):
    s, pad = weight._gdl_stem
    n, c, k, _ = weight.shape
    b, e = STEM_BLOCK, STEM_BLOCK // s
    lo, hi = -pad, s * (e - 1) - pad + k - 1            # pixel offsets (relative to block 4 B) a filter can touch
    if lo < -b or hi > 2 * b - 1:
        raise NotImplementedError(f"stem {k}x{k} / stride {s} / pad {pad} reaches beyond the neighbouring 4-pixel blocks")
    dev = weight.device
    eps = torch.arange(e, device=dev).view(e, 1, 1)
    jj = torch.arange(-1, 2, device=dev).view(1, 3, 1)
    dd = torch.arange(b, device=dev).view(1, 1, b)
    r = b * jj + dd - s * eps + pad                                                   # [e, 3, 4]
    ok = (r >= 0) & (r < k)
    r = r.clamp(0, k - 1)
    cc = torch.arange(c, device=dev).view(1, 1, 1, 1, c, 1, 1)
    ry, rx = r.view(e, 1, 3, 1, 1, b, 1), r.view(1, e, 1, 3, 1, 1, b)
    idx = (cc * k + ry) * k + rx                                                      # [ey, ex, jy, jx, c, dy, dx]
    valid = (ok.view(e, 1, 3, 1, 1, b, 1) & ok.view(1, e, 1, 3, 1, 1, b)).expand_as(idx)
    return e, idx.reshape(e * e, 9, c * b * b), valid.reshape(e * e, 9, c * b * b)


def stem_operands(weight: Tensor, cd: torch.dtype, cpad: int, npad: int) -> Tensor:
    """[e*e, Npad, 9 * Cpad] phase filters of a marked stem in the compute dtype (see mark_stem)."""
    def build():
        e, idx, valid = _stem_index(weight)
        n = weight.shape[0]
        flat = weight.detach().reshape(n, -1).float()
        m = torch.zeros((e * e, npad, 9, cpad), device=weight.device, dtype=torch.float32)
        m[:, :n, :, : idx.shape[-1]] = torch.where(valid.unsqueeze(1), flat[:, idx].permute(1, 0, 2, 3), flat.new_zeros(()))
        m = m.view(e * e, npad, 9 * cpad)
        return m if cd == torch.float32 else ops.cast(m, cd)
    return cached((weight,), f"stemw:{cd}:{cpad}:{npad}", build)


def stem_conv(x: Tensor, weight: Tensor, npad: int, **epilogue) -> Tensor:
    """The stem convolution on the space-to-depth image x [B, H/4, W/4, Cpad] -> [B, H/s, W/s, Npad]: e*e phase convolutions,
    each writing its strided slice of the output (``epilogue``: gdlhip.ops.conv_gemm's scale / shift / act keywords)."""
    s, _ = weight._gdl_stem
    e = STEM_BLOCK // s
    wq = stem_operands(weight, x.dtype, x.shape[-1], npad)
    b, hb, wb, _ = x.shape
    y = torch.empty((b, hb * e, wb * e, npad), device=x.device, dtype=x.dtype)
    for ph in range(e * e):
        ops.conv_gemm(x, wq[ph], R=3, S=3, pad=1, out=y[:, ph // e::e, ph % e::e, :], **epilogue)
    return y


def stem_param_grad(x: Tensor, dy: Tensor, weight: Tensor) -> Tensor:
    """dL/dW of a marked stem from the space-to-depth image and the dense output gradient: one nine-tap weight gradient per
    phase on the strided view of dy, scattered back to the parameter's k x k taps (every tap has one slot per phase)."""
    e, idx, _ = _stem_index(weight)
    slots, taps = _stem_lists(weight)
    n, cpad = weight.shape[0], x.shape[-1]
    out = torch.zeros((n, weight[0].numel()), device=x.device, dtype=torch.float32)
    for ph in range(e * e):
        dw = ops.conv_wgrad(x, dy[:, ph // e::e, ph % e::e, :], R=3, S=3, pad=1).view(-1, 9, cpad)[:n, :, : idx.shape[-1]]
        out.index_add_(1, taps[ph], dw.reshape(n, -1).index_select(1, slots[ph]))     # (each tap has one slot per phase)
    return out.reshape(weight.shape)


def space_to_depth_image(img: Tensor, cd: torch.dtype) -> Tensor:
    """NCHW f32 image -> [B, H/4, W/4, Cpad] in the compute dtype, channel order (c, dy, dx), Cpad = 16 C rounded up to a
    power of two >= 16 (whole or evenly divided K chunks for the implicit-GEMM kernels)."""
    b, c, h, w = img.shape
    if h % STEM_BLOCK or w % STEM_BLOCK:
        raise ValueError(f"image {h}x{w}: height and width must be multiples of {STEM_BLOCK}")
    cpad = 16
    while cpad < c * STEM_BLOCK * STEM_BLOCK:
        cpad *= 2
    hb, wb = h // STEM_BLOCK, w // STEM_BLOCK
    return ops.patchify(img, STEM_BLOCK, 0, hb, wb, cpad, cd, stride=STEM_BLOCK).view(b, hb, wb, cpad)


def _is_flat(weight: Tensor) -> bool:
    return weight.dim() == 2 or getattr(weight, "_gdl_flat", False)


def _wshape(weight: Tensor) -> tuple[int, int, int, int]:
    """(N, C, R, S) of the DENSE filter the kernels run (C = all input channels, also for a grouped parameter)."""
    if _is_flat(weight):
        return weight.shape[0], weight[0].numel(), 1, 1
    n, c, r, s = weight.shape
    if hasattr(weight, "_gdl_stem"):
        return n, c * STEM_BLOCK * STEM_BLOCK, 3, 3
    return n, c * _groups(weight), r, s


def _matrix3(weight: Tensor) -> Tensor:
    """f32 [N, R*S, C] view/copy of a conv parameter ([N,C,R,S]), a 2-D [N,K] matrix or a flat stem (taps = 1); a grouped
    parameter becomes its block-diagonal dense filter."""
    if _is_flat(weight):
        return weight.detach().reshape(weight.shape[0], 1, -1)
    n, c, r, s = weight.shape
    if hasattr(weight, "_gdl_stem"):
        raise ValueError("a marked stem has one filter per sub-pixel phase: use stem_operands / stem_conv")
    m = conv_weight_matrix(weight).view(n, r * s, c)
    g = _groups(weight)
    if g == 1:
        return m
    dense = torch.zeros((g, n // g, r * s, g, c), device=weight.device, dtype=m.dtype)
    idx = torch.arange(g, device=weight.device)
    dense[idx, :, :, idx, :] = m.view(g, n // g, r * s, c)          # group i: outputs i*N/g.., inputs i*c..
    return dense.view(n, r * s, g * c)


def padded_operands(weight: Tensor, cd: torch.dtype, cpad: int, npad: int) -> tuple[Tensor, Tensor]:
    """(forward operand [Npad, R*S*Cpad], data-gradient operand [Cpad, (R*S flipped)*Npad]) in the compute dtype."""
    def build():
        n, c, r, s = _wshape(weight)
        m = torch.zeros((npad, r * s, cpad), device=weight.device, dtype=torch.float32)
        m[:n, :, :c] = _matrix3(weight)
        m = m.view(npad, r * s * cpad)
        fwd = m if cd == torch.float32 else ops.cast(m, cd)
        return fwd, ops.pack_dgrad(m, npad, r * s, cpad, cd)
    return cached((weight,), f"padw:{cd}:{cpad}:{npad}", build)


def padded_t_operand(weight: Tensor, cd: torch.dtype, cpad: int, npad: int) -> Tensor:
    """[(R*S*Cpad), Npad] operand for the GEMM + col2im data gradient of a strided conv."""
    def build():
        n, c, r, s = _wshape(weight)
        m = torch.zeros((npad, r * s, cpad), device=weight.device, dtype=torch.float32)
        m[:n, :, :c] = _matrix3(weight)
        return ops.pack_dgrad(m.view(npad, r * s * cpad), npad, 1, r * s * cpad, cd)
    return cached((weight,), f"padwT:{cd}:{cpad}:{npad}", build)


def _padvec(v: Tensor, npad: int, fill: float = 0.0) -> Tensor:
    if v.numel() == npad:
        return v.detach()
    out = torch.full((npad,), fill, device=v.device, dtype=torch.float32)
    out[: v.numel()] = v.detach()
    return out


def _param_grad(dw: Tensor, weight: Tensor, cpad: int) -> Tensor:
    """[Npad, R*S*Cpad] f32 weight gradient -> the parameter's logical shape (real channels only)."""
    n, c, r, s = _wshape(weight)
    d = dw.view(-1, r * s, cpad)[:n, :, :c]
    if _is_flat(weight):
        return d.reshape(weight.shape)
    g = _groups(weight)
    if g > 1:      # the parameter's gradient = the block diagonal of the dense filter's
        idx = torch.arange(g, device=dw.device)
        d = d.reshape(g, n // g, r * s, g, c // g)[idx, :, :, idx, :].reshape(n, r * s, c // g)
        c = c // g
    if r == 1 and s == 1:
        return d.reshape(n, c, 1, 1)
    return d.reshape(n, r, s, c).permute(0, 3, 1, 2)


def _conv_dx(dy: Tensor, weight: Tensor, cd: torch.dtype, cpad: int, npad: int, stride: int, pad: int,
             in_hw: tuple[int, int]) -> Tensor:
    n, c, r, s = _wshape(weight)
    if stride == 1:
        return ops.conv_gemm(dy, padded_operands(weight, cd, cpad, npad)[1], R=r, S=s, pad=r - 1 - pad)
    b, ho, wo, _ = dy.shape
    cols = ops.linear(dy.reshape(-1, npad), padded_t_operand(weight, cd, cpad, npad), None)
    return ops.col2im(cols, b, ho, wo, r, s, cpad, stride, pad, in_hw[0], in_hw[1], cd)


# ------------------------------------------------------------------ conv -> BN(batch stats) -> [ReLU]
class _ConvBNTrain(Function):
    """Training-mode Conv2d(bias=False) + BatchNorm2d / SyncBatchNorm + optional ReLU with stride and channel
    padding (torchvision resnet.py conv-bn-relu; smp base/modules.py Conv2dReLU)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, momentum, eps, stride, pad, relu, sync_group):
        cd = x.dtype
        n, c, r, s = _wshape(weight)
        cpad, npad = x.shape[-1], pad_to(n, grain(cd))
        if cpad < c or cpad % grain(cd):
            raise ValueError(f"conv input has {cpad} channels, weight expects {c} (padded to a multiple of {grain(cd)})")
        sg = supergroups(weight, cpad, npad, cd)
        if hasattr(weight, "_gdl_stem"):
            y = stem_conv(x, weight, npad)
        elif sg is not None:
            y = ops.conv_gemm_grouped(x, grouped_operands(weight, cd, sg)[0], R=r, S=s, stride=stride, pad=pad)
        else:
            wq, _ = padded_operands(weight, cd, cpad, npad)
            y = ops.conv_gemm(x, wq, R=r, S=s, stride=stride, pad=pad)
        world = _world(sync_group) if sync_group is not False else 1
        p_local, p_share, total = y.numel() // npad, None, y.numel() // npad
        in_kernel = world == 1 and running_mean is not None and npad == n   # the statistics kernel updates the buffers itself
        mean, var = ops.bn_stats(y, running_mean, running_var, momentum) if in_kernel else ops.bn_stats(y)
        if world > 1:
            mean, var, total = sync_batch_stats(mean, var, sync_group or None, count=p_local)
            p_share = p_local / total
        if in_kernel:     # written through raw pointers: invalidate the eval-mode fold cache
            mark_updated(running_mean)
            mark_updated(running_var)
        elif running_mean is not None:
            update_running_stats(running_mean, running_var, mean[:n], var[:n], momentum, total)
        g, b = _padvec(gamma, npad), _padvec(beta, npad)
        out = ops.bn_apply(y, mean, var, g, b, eps, relu)
        ctx.save_for_backward(x, weight, y, mean, var, g, b)
        ctx.cfg = (stride, pad, relu, eps, sync_group, world, cpad, npad, p_local, p_share)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, y, mean, var, g, b = ctx.saved_tensors
        stride, pad, relu, eps, sync_group, world, cpad, npad, p_local, p_share = ctx.cfg
        n, c, r, s = _wshape(weight)
        gout = _dense(gout if gout.dtype == y.dtype else to_compute(gout, y.dtype))
        dgamma, dbeta = ops.bn_bwd_reduce(y, gout, mean, var, g, b, eps, relu)
        sg, sb = dgamma, dbeta
        if world > 1:
            sg, sb = sync_sum_pair(dgamma, dbeta, sync_group or None)
            sg, sb = sg * p_share, sb * p_share      # see gdlhip.nn._ConvBNActTrain.backward
        dy = ops.bn_bwd_dx(y, gout, mean, var, g, b, eps, relu, sg, sb, p_local, out=y)
        dw = None
        stem = hasattr(weight, "_gdl_stem")
        sg = supergroups(weight, cpad, npad, x.dtype)
        if ctx.needs_input_grad[1]:
            if sg is not None:
                dw = _grouped_param_grad(ops.conv_wgrad_grouped(x, dy, Z=sg[0], R=r, S=s, stride=stride, pad=pad), weight, sg)
            else:
                dw = (stem_param_grad(x, dy, weight) if stem else
                      _param_grad(ops.conv_wgrad(x, dy, R=r, S=s, stride=stride, pad=pad), weight, cpad))
        dx = None
        if ctx.needs_input_grad[0]:
            if stem:
                raise NotImplementedError("gdlhip: no gradient with respect to the raw image of a stem convolution")
            if sg is not None and stride == 1:
                dx = ops.conv_gemm_grouped(dy, grouped_operands(weight, x.dtype, sg)[1], R=r, S=s, pad=r - 1 - pad)
            else:
                dx = _conv_dx(dy, weight, x.dtype, cpad, npad, stride, pad, (x.shape[1], x.shape[2]))
        return dx, dw, dgamma[:n], dbeta[:n], None, None, None, None, None, None, None, None


def conv_bn(x: Tensor, weight: Tensor, norm: nn.Module, *, stride: int = 1, pad: int = 0, relu: bool = True,
            resid: Tensor | None = None) -> Tensor:
    """conv(bias=False) -> BN -> [+ resid] -> [ReLU] on NHWC (compute dtype); output channels padded to the
    K-chunk.  Training: batch statistics (autograd node); eval: BN folded into the GEMM epilogue."""
    cd = x.dtype
    n, c, r, s = _wshape(weight)
    if norm.training:
        sync_group = norm.process_group if isinstance(norm, nn.SyncBatchNorm) else False
        momentum = 0.1 if norm.momentum is None else norm.momentum
        y = _ConvBNTrain.apply(x, weight, norm.weight, norm.bias, norm.running_mean, norm.running_var, momentum,
                               norm.eps, stride, pad, relu and resid is None, sync_group)
        gnn.bump(norm.num_batches_tracked)
        if resid is not None:
            y = add_relu(y, resid) if relu else y + resid
        return y
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        msg = ("gdlhip: autograd through eval-mode BatchNorm is not implemented; call under torch.no_grad() for "
               "inference or model.train() for training")
        raise NotImplementedError(msg)
    cpad, npad = x.shape[-1], pad_to(n, grain(cd))

    def fold():
        scale, shift = ops.bn_fold(norm.weight.detach(), norm.bias.detach(), norm.running_mean, norm.running_var,
                                   norm.eps)
        return _padvec(scale, npad), _padvec(shift, npad)
    scale, shift = cached((norm.weight, norm.bias, norm.running_mean, norm.running_var), f"bnfold:{npad}", fold)
    act = ACT_NONE if not relu else (ACT_RELU if resid is None else ACT_RESID_RELU)
    sg = supergroups(weight, cpad, npad, cd)
    if sg is not None and resid is None:
        # batched grouped convolution (no epilogue vectors per z), then the folded BatchNorm + ReLU as one pass
        y = ops.conv_gemm_grouped(x, grouped_operands(weight, cd, sg)[0], R=r, S=s, stride=stride, pad=pad)
        return ops.bn_apply(y, norm.running_mean, norm.running_var, norm.weight.detach(), norm.bias.detach(), norm.eps, relu)
    if hasattr(weight, "_gdl_stem"):
        if resid is not None:
            raise ValueError("a stem convolution takes no residual operand")
        return stem_conv(x, weight, npad, scale=scale, shift=shift, act=act)
    return ops.conv_gemm(x, padded_operands(weight, cd, cpad, npad)[0], R=r, S=s, stride=stride, pad=pad, scale=scale,
                         shift=shift, act=act, resid=resid)


# ------------------------------------------------------------------ plain conv + bias (segmentation head)
class _ConvBias(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad, out_dtype):
        cd = x.dtype
        n, c, r, s = _wshape(weight)
        cpad = x.shape[-1]
        wq = padded_operands(weight, cd, cpad, n)[0]
        ctx.save_for_backward(x, weight)
        ctx.cfg = (pad, cpad)
        return ops.conv_gemm(x, wq, R=r, S=s, pad=pad, bias=bias.detach(), out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        pad, cpad = ctx.cfg
        cd = x.dtype
        n, c, r, s = _wshape(weight)
        g = g.contiguous()                                  # dense NHWC [B,H,W,n] f32 (n = classes, any count)
        npad = pad_to(n, 8)                                 # 16-byte rows for the weight-gradient kernel
        dy = ops.pad_channels(g, npad, cd)
        dw = _param_grad(ops.conv_wgrad(x, dy, R=r, S=s, pad=pad), weight, cpad) if ctx.needs_input_grad[1] else None
        db = ops.colsum(dy)[:n] if ctx.needs_input_grad[2] else None
        dx = None
        if ctx.needs_input_grad[0]:
            npk = pad_to(n, grain(cd))                      # K of the data-gradient GEMM
            dyk = dy if npk == npad else ops.pad_channels(g, npk, cd)
            dx = ops.conv_gemm(dyk, padded_operands(weight, cd, cpad, npk)[1], R=r, S=s, pad=r - 1 - pad)
        return dx, dw, db, None, None


def conv_bias(x: Tensor, conv: nn.Conv2d, out_dtype: torch.dtype = torch.float32) -> Tensor:
    """Conv2d(+bias) with few output channels (the 3x3 classifier): NHWC in, NHWC [B,H,W,N] out."""
    return _ConvBias.apply(x, conv.weight, conv.bias, conv.padding[0], out_dtype)


# ------------------------------------------------------------------ pooling / resampling / residual
class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool3x3s2(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.maxpool3x3s2_bwd(x, _dense(g))


def maxpool3x3s2(x: Tensor) -> Tensor:
    return _MaxPool.apply(x)


class _UpCat(Function):
    """cat([nearest2x(x), *skips], channel dim) written straight into one NHWC buffer
    (smp decoders/unetplusplus/decoder.py DecoderBlock.forward + the dense-skip torch.cat)."""

    @staticmethod
    def forward(ctx, x, *skips):
        b, h, w, c = x.shape
        chans = [c] + [s.shape[3] for s in skips]
        out = torch.empty((b, 2 * h, 2 * w, sum(chans)), device=x.device, dtype=x.dtype)
        ops.nearest2x(x, out=out[..., :c])
        off = c
        for s in skips:
            ops.copy_cast(s, out=out[..., off:off + s.shape[3]])
            off += s.shape[3]
        ctx.chans = chans
        return out

    @staticmethod
    def backward(ctx, g):
        chans = ctx.chans
        grads = [ops.nearest2x_bwd(g[..., : chans[0]]) if ctx.needs_input_grad[0] else None]
        off = chans[0]
        for i, c in enumerate(chans[1:]):
            grads.append(g[..., off:off + c] if ctx.needs_input_grad[i + 1] else None)
            off += c
        return tuple(grads)


def up_cat(x: Tensor, skips: list[Tensor]) -> Tensor:
    return _UpCat.apply(x, *skips)


class _AddRelu(Function):
    @staticmethod
    def forward(ctx, a, b):
        out = ops.add_relu(a, b)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        d = ops.relu_bwd(out, _dense(g))
        return d, d


def add_relu(a: Tensor, b: Tensor) -> Tensor:
    return _AddRelu.apply(a.contiguous(), b.contiguous())


class _LogitsNCHW(Function):
    """NHWC f32 logits [B,H,W,K] -> NCHW f32 (what the loss / argmax kernels consume)."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.upsample_logits(x, ctx.hw)

    @staticmethod
    def backward(ctx, g):
        return ops.upsample_logits_bwd(g.contiguous(), ctx.hw)


def logits_nchw(x: Tensor) -> Tensor:
    return _LogitsNCHW.apply(x)
