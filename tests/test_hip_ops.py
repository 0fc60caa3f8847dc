"""GPU parity tests, op level: every HIP kernel (through the C-ABI) vs a plain PyTorch f32/f64
CPU reference of the same op on seeded inputs.  f32 tolerance 1e-4 (relative to the output
scale), bf16 tolerance 2e-2."""

import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

gdlhip = pytest.importorskip("gdlhip")
from gdlhip import nn as gnn  # noqa: E402
from gdlhip import ops  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 2e-2


def close(got, ref, dtype, what="", scale=None):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    s = ref.abs().max().item() if scale is None else scale
    err = (got - ref).abs().max().item()
    assert err <= tol(dtype) * max(s, 1e-6), f"{what}: max err {err:.3e} vs scale {s:.3e}"


def rnd(*shape, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g).to(dtype)


def q(t, dtype):
    """quantise a f32 CPU tensor to the compute dtype (reference sees the same operand values)."""
    return t.to(dtype).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,N", [(300, 128, 192), (64, 64, 64), (1297, 768, 256), (3, 128, 1024), (500, 32, 128),
                                   (2000, 160, 640), (77, 24, 8)])
def test_linear_plain(dtype, M, K, N):
    x, w = q(rnd(M, K), dtype), q(rnd(N, K, seed=1), dtype)
    y = ops.linear(x.to(DEV, dtype), w.to(DEV, dtype), out_dtype=torch.float32)
    close(y, x @ w.t(), dtype, "linear")


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_epilogues(dtype):
    B, T, K, N = 2, 150, 128, 256
    x, w = q(rnd(B, T, K), dtype), q(rnd(N, K, seed=1), dtype) * 0.1
    bias, scale, shift = rnd(N, seed=2), rnd(N, seed=3), rnd(N, seed=4)
    resid, bs = rnd(B, T, N, seed=5), torch.tensor([0.0, 1.25])
    xd, wd = x.to(DEV, dtype), w.to(DEV, dtype)
    base = x @ w.t()
    y = ops.linear(xd, wd, bias.to(DEV), act=ops.ACT_GELU, out_dtype=torch.float32)
    close(y, F.gelu(base + bias), dtype, "gelu")
    y = ops.linear(xd, wd, bias.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), act=ops.ACT_RELU,
                   out_dtype=torch.float32)
    close(y, F.relu((base + bias) * scale + shift), dtype, "bn-fold relu")
    out = torch.empty(B, 1, T, N, device=DEV)
    ops.conv_gemm(xd.view(B, 1, T, K), wd, bias=bias.to(DEV), scale=scale.to(DEV), batch_scale=bs.to(DEV),
                  resid=resid.to(DEV).view(B, 1, T, N), out=out)
    ref = resid + ((base + bias) * scale) * bs.view(B, 1, 1)
    close(out.view(B, T, N), ref, dtype, "layerscale+droppath+residual")
    y = ops.linear(xd, wd, out_dtype=dtype, alpha=0.5)
    close(y, 0.5 * base, dtype if dtype == torch.float32 else torch.bfloat16, "alpha / out dtype")


@pytest.mark.parametrize("variant", [1, 3, 6, 8, 9])
@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_epilogue_row_segments(dtype, variant):
    """The coalesced (LDS-transposed) epilogue of the 128^2, 256^2 and dual-resident 256 x 128 (variant 6) tiles: residual in f32 / bf16, DropPath scale with
    a sample boundary inside a 32-row pass (T = 1297), bf16 / f32 outputs, pre-activation copy, GELU-gradient multiply,
    M and N tails, channel-slice output (row stride > N)."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    if variant in (6, 8, 9) and dtype != torch.bfloat16:
        pytest.skip("the dual-resident, one-wave-per-SIMD and persistent tiles are bf16 kernels")
    B, T, K, N = 3, 1297, 128, 208          # M = 3891 (tail of 51 rows), N % 64 = 16, N % 256 != 0
    if variant in (3, 8, 9):
        N = 512                              # the 256^2 tiles need N % 256 == 0
    x, w = q(rnd(B, T, K), dtype), q(rnd(N, K, seed=1), dtype) * 0.1
    bias, scale, shift = rnd(N, seed=2), rnd(N, seed=3), rnd(N, seed=4)
    resid, bs = rnd(B, T, N, seed=5), torch.tensor([0.0, 1.25, 0.5])
    xd, wd = x.to(DEV, dtype).view(B, 1, T, K), w.to(DEV, dtype)
    base = (x @ w.t())
    lib.gdl_debug_force_conv_variant(variant)
    try:
        for odt in DTYPES:
            for rdt in DTYPES:
                out = torch.empty(B, 1, T, N, device=DEV, dtype=odt)
                rq = q(resid, rdt)
                ops.conv_gemm(xd, wd, bias=bias.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), batch_scale=bs.to(DEV),
                              resid=rq.to(DEV, rdt).view(B, 1, T, N), out=out)
                ref = rq + ((base + bias) * scale + shift) * bs.view(B, 1, 1)
                close(out.view(B, T, N), ref, dtype if odt == torch.float32 else torch.bfloat16, f"resid {odt} {rdt}")
        # plain bf16 output (packed rows) into a channel slice of a wider buffer
        wide = torch.zeros(B, 1, T, N + 64, device=DEV, dtype=torch.bfloat16)
        ops.conv_gemm(xd, wd, bias=bias.to(DEV), act=ops.ACT_RELU, out=wide[..., 32:32 + N])
        close(wide[..., 32:32 + N].reshape(B, T, N), F.relu(base + bias), torch.bfloat16, "slice out")
        assert float(wide[..., :32].abs().max()) == 0.0 and float(wide[..., 32 + N:].abs().max()) == 0.0
        if variant == 1:                     # training-only epilogue features live in the small-tile instantiations
            pre = torch.empty(B, 1, T, N, device=DEV, dtype=dtype)
            y = ops.conv_gemm(xd, wd, bias=bias.to(DEV), act=ops.ACT_GELU, aux_out=pre, out_dtype=dtype)
            close(pre.view(B, T, N), base + bias, dtype, "aux_out")
            close(y.view(B, T, N), F.gelu(base + bias), dtype, "gelu with aux_out")
            u = rnd(B, T, N, seed=9)
            g = ops.conv_gemm(xd, wd, act=ops.ACT_MUL_GELU_GRAD, resid=u.to(DEV).view(B, 1, T, N), out_dtype=torch.float32)
            ug = u.clone().requires_grad_(True)
            F.gelu(ug).sum().backward()
            close(g.view(B, T, N), base * ug.grad, dtype, "gelu-grad multiply")
    finally:
        lib.gdl_debug_force_conv_variant(-1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,N", [(2, 9, 9, 64, 256), (1, 36, 36, 128, 256), (3, 5, 7, 64, 64), (1, 2, 2, 64, 128)])
def test_up4_conv3x3_matches_upsample_then_conv(dtype, B, H, W, C, N):
    """The fused sub-pixel form of conv3x3(pad 1)(bilinear x4 (align_corners=False)) (multilevel_neck.py:157-158) vs
    torch's two ops, every pixel including the border lines the convolution's zero padding touches; with the folded-BN
    + ReLU epilogue of the eval path; and through the training ConvModule node against the unfused path."""
    x, w = q(rnd(B, H, W, C), dtype), q(rnd(N, C, 3, 3, seed=1) * 0.1, dtype)
    bias, scale, shift = rnd(N, seed=2), rnd(N, seed=3), rnd(N, seed=4)
    up = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=4, mode="bilinear", align_corners=False)
    ref = F.conv2d(up, w, bias, padding=1).permute(0, 2, 3, 1)
    wm = w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    ws = ops.subpix4_weights(wm.to(DEV), C, dtype)
    y = ops.up4_conv3x3(x.to(DEV, dtype), ws, bias=bias.to(DEV))
    assert y.shape == (B, 4 * H, 4 * W, N)
    # bf16: the reference rounds the upsampled map to bf16, the fused form rounds the combined weights instead
    close(y, ref, dtype, "up4 conv", scale=None if dtype == torch.float32 else 2 * ref.abs().max().item())
    for name, sl in (("top", (slice(None), 0)), ("bottom", (slice(None), -1)), ("left", (slice(None), slice(None), 0)),
                     ("right", (slice(None), slice(None), -1))):
        close(y[sl], ref[sl], dtype, f"up4 conv {name} line", scale=ref.abs().max().item() * (1 if dtype == torch.float32 else 2))
    y = ops.up4_conv3x3(x.to(DEV, dtype), ws, bias=bias.to(DEV), scale=scale.to(DEV), shift=shift.to(DEV), act=ops.ACT_RELU)
    close(y, F.relu(ref * scale + shift), dtype, "up4 conv + folded BN + ReLU",
          scale=None if dtype == torch.float32 else 2 * ref.abs().max().item())
    if B < 2:
        return
    # training node: same gradients as the unfused path (the backward runs on the recomputed upsampled map)
    conv = torch.nn.Conv2d(C, N, 3, padding=1).to(DEV).to(memory_format=torch.channels_last)
    norm = torch.nn.BatchNorm2d(N).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(w.to(DEV))
    got = {}
    for fuse in (True, False):
        gnn.FUSE_UP4 = fuse
        try:
            for p in list(conv.parameters()) + list(norm.parameters()):
                p.grad = None
            norm.reset_running_stats()
            xd = x.to(DEV, dtype).requires_grad_()
            out = gnn.conv_bn_act(xd, conv, norm.train(), relu=True, up4=True)
            out.float().square().mean().backward()
            got[fuse] = (out.detach().float(), xd.grad.float(), conv.weight.grad.float().clone(), norm.weight.grad.clone(),
                         norm.running_var.clone())
        finally:
            gnn.FUSE_UP4 = True
    tol_dt = dtype
    for a, b, what in zip(got[True], got[False], ("out", "dx", "dw", "dgamma", "running_var")):
        close(a, b, tol_dt, f"fused vs unfused {what}", scale=None if dtype == torch.float32 else 4 * b.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,N,up", [(2, 9, 9, 64, 256, 2), (1, 12, 7, 128, 64, 2), (3, 5, 6, 64, 64, 4), (2, 36, 36, 64, 64, 2)])
def test_resized_conv_backward_at_low_resolution(dtype, B, H, W, C, N, up):
    """ConvModule on a bilinearly upsampled input (MultiLevelNeck's x2 / x4 levels, multilevel_neck.py:56-67,157-158): the
    node's data and weight gradients are GEMMs over the LOW-resolution pixels (nine gathered maps, ops.resize_conv3x3_bwd);
    vs torch autograd of interpolate -> conv2d -> batch_norm -> relu on the CPU, every border pixel included."""
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, 3, 3, seed=1) * 0.1, dtype)
    conv_r = torch.nn.Conv2d(C, N, 3, padding=1, bias=False)
    bn_r = torch.nn.BatchNorm2d(N)
    with torch.no_grad():
        conv_r.weight.copy_(w)
        bn_r.weight.copy_(rnd(N, seed=2).abs() + 0.5)
        bn_r.bias.copy_(rnd(N, seed=3) * 0.1)
    xr = x.clone().requires_grad_()
    yr = F.relu(bn_r(conv_r(F.interpolate(xr, scale_factor=up, mode="bilinear", align_corners=False))))
    gy = q(rnd(*yr.shape, seed=5), dtype)
    yr.backward(gy)
    import copy
    conv = copy.deepcopy(conv_r).to(DEV).to(memory_format=torch.channels_last)
    norm = copy.deepcopy(bn_r).to(DEV)
    norm.reset_running_stats()
    for p in list(conv.parameters()) + list(norm.parameters()):
        p.grad = None
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_()
    y = gnn.conv_bn_act(xd, conv, norm.train(), relu=True, up=up)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype))
    close(y.permute(0, 3, 1, 2), yr, dtype, "output")
    # bf16: conv output, dy and the nine gathered maps are stored in bf16 -- gradients are held in the L2 norm (3-4 % is
    # what train-mode BatchNorm's backward costs in bf16 on maps this small, fused or not: test_pyramid_fuse_* prints it)
    for got, ref, what in ((xd.grad.permute(0, 3, 1, 2), xr.grad, "dx"), (conv.weight.grad, conv_r.weight.grad, "dw"),
                           (norm.weight.grad, bn_r.weight.grad, "dgamma"), (norm.bias.grad, bn_r.bias.grad, "dbeta")):
        got, ref = got.float().cpu(), ref.float()
        if dtype == torch.float32:
            close(got, ref, dtype, what, scale=ref.abs().max().item())
        else:
            rel = float((got - ref).norm() / ref.norm())
            assert rel <= 6e-2, (what, rel)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,N,R", [(2, 10, 12, 64, 96, 3), (1, 18, 18, 128, 256, 3), (2, 7, 5, 64, 64, 1),
                                         (3, 36, 36, 64, 128, 3),
                                         # channel tails (C not a multiple of the K chunk: MiT-B0's 32 / 160 channels)
                                         (2, 9, 11, 32, 64, 3), (1, 40, 40, 160, 256, 1), (2, 6, 6, 96, 32, 3),
                                         (1, 48, 48, 40, 128, 1), (3, 20, 20, 8, 16, 3)])
def test_conv_nhwc(dtype, B, H, W, C, N, R):
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, R, R, seed=1) * 0.1, dtype)
    bias = rnd(N, seed=2)
    ref = F.conv2d(x, w, bias, padding=R // 2)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wq = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV, dtype)
    y = ops.conv_gemm(xn, wq, R=R, S=R, pad=R // 2, bias=bias.to(DEV), out_dtype=torch.float32)
    close(y.permute(0, 3, 1, 2), ref, dtype, "conv")
    # output in the compute dtype inside a wider buffer: the 16-byte store path must respect N (also N % 32 == 16)
    buf = torch.full((B, H, W, N + 16), 7.0, device=DEV, dtype=dtype)
    ops.conv_gemm(xn, wq, R=R, S=R, pad=R // 2, bias=bias.to(DEV), out=buf[..., 8:8 + N])
    close(buf[..., 8:8 + N].float().permute(0, 3, 1, 2), ref, dtype, "conv into a slice")
    assert (buf[..., :8] == 7).all() and (buf[..., 8 + N:] == 7).all()
    dense = ops.conv_gemm(xn, wq, R=R, S=R, pad=R // 2, bias=bias.to(DEV), out_dtype=dtype)
    close(dense.float().permute(0, 3, 1, 2), ref, dtype, "conv dense compute-dtype output")


def test_conv_strided_views():
    """channel-sliced input and output buffers (concat-free decoder), tokens with a cls row."""
    dtype = torch.float32
    B, H, W = 2, 6, 6
    big_in = rnd(B, H, W, 160).to(DEV)
    x = big_in[..., 32:96]                     # 64 channels inside a 160-channel buffer
    w = rnd(64, 64 * 9, seed=1).to(DEV) * 0.1
    big_out = torch.zeros(B, H, W, 192, device=DEV)
    ops.conv_gemm(x, w, R=3, S=3, pad=1, out=big_out[..., 64:128])
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu(), w.cpu().view(64, 3, 3, 64).permute(0, 3, 1, 2), padding=1)
    close(big_out[..., 64:128].permute(0, 3, 1, 2), ref, dtype, "sliced conv")
    assert big_out[..., :64].abs().max().item() == 0 and big_out[..., 128:].abs().max().item() == 0
    tok = rnd(B, 37, 64).to(DEV)               # taps: drop the cls token, view as 6x6 NHWC
    tap = tok[:, 1:, :].unflatten(1, (6, 6))
    y = ops.conv_gemm(tap, w[:, :64].contiguous(), out_dtype=torch.float32)
    close(y.reshape(B, 36, 64), tok[:, 1:, :].cpu() @ w[:, :64].cpu().t(), dtype, "tap 1x1")


def test_conv_arg_validation():
    x = torch.zeros(1, 4, 4, 50, device=DEV)
    w = torch.zeros(64, 50, device=DEV)
    with pytest.raises(ValueError):
        ops.conv_gemm(x, w)  # C not a multiple of 4 (16-byte pieces)
    with pytest.raises(ValueError):
        ops.conv_gemm(torch.zeros(1, 4, 4, 64), torch.zeros(64, 64))  # CPU tensors


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,H,hd", [(2, 70, 2, 64), (1, 197, 3, 64), (1, 132, 4, 32)])
def test_attention_unfused(dtype, B, N, H, hd):
    D = H * hd
    qkv = q(rnd(B, N, 3 * D), dtype)
    o = ops.attention_unfused(*ops.split_qkv(qkv.to(DEV, dtype)), H)
    qq, kk, vv = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, N, D)
    close(o, ref, dtype, "attention_unfused")


@pytest.mark.parametrize("B,N,H", [(2, 70, 2), (1, 1297, 2), (2, 128, 3), (1, 65, 1)])
def test_attention_flash(B, N, H):
    dtype, hd = torch.bfloat16, 64
    D = H * hd
    qkv = q(rnd(B, N, 3 * D) * 1.5, dtype)
    o = ops.attention_flash(*ops.split_qkv(qkv.to(DEV, dtype)), H)
    qq, kk, vv = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, N, D)
    close(o, ref, dtype, "attention_flash")


@pytest.mark.parametrize("B,Nq,Nkv,H", [(2, 70, 70, 2), (2, 300, 300, 4), (1, 1297, 1297, 8), (2, 333, 100, 4), (1, 64, 257, 3),
                                        # few keys, many queries (MiT stage 1/2): the dK/dV kernel splits the query range
                                        (2, 4096, 256, 1), (1, 1000, 100, 2)])
def test_attention_flash_lse_and_fused_backward(B, Nq, Nkv, H):
    """Second-generation attention kernels: LSE output, V read row-major, and the fused backward (probabilities
    recomputed from LSE, no [Nq, Nkv] tensor) vs torch autograd; packed / strided operands; both block orders
    (B*H a multiple of 8 or not); separate query / key lengths (MiT spatial reduction); the v1 kernel as a cross-check."""
    dtype, hd = torch.bfloat16, 64
    D = H * hd
    qf = q(rnd(B, Nq, D) * 1.2, dtype).requires_grad_()
    kvf = q(rnd(B, Nkv, 2 * D, seed=1) * 1.2, dtype).requires_grad_()
    do = q(rnd(B, Nq, D, seed=2), dtype)

    def heads(t, n):
        return t.reshape(B, n, H, hd).transpose(1, 2)
    sc = heads(qf, Nq) @ heads(kvf[..., :D], Nkv).transpose(-1, -2) * hd ** -0.5
    out = (sc.softmax(-1) @ heads(kvf[..., D:], Nkv)).transpose(1, 2).reshape(B, Nq, D)
    out.backward(do)
    qd, kvd = qf.detach().to(DEV, dtype), kvf.detach().to(DEV, dtype)
    o, lse = ops.attention_flash(qd, kvd[..., :D], kvd[..., D:], H, return_lse=True)
    close(o, out, dtype, "flash fwd2")
    close(lse, torch.logsumexp(sc.detach(), -1), torch.float32, "lse", scale=max(1.0, sc.detach().abs().max().item()) * 20)
    close(ops.attention_flash_v1(qd, kvd[..., :D], kvd[..., D:], H), out, dtype, "flash v1")
    dq = torch.empty_like(qd)
    dkv = torch.empty_like(kvd)
    ops.attention_bwd(qd, kvd[..., :D], kvd[..., D:], do.to(DEV, dtype), H, dq, dkv[..., :D], dkv[..., D:], o=o, lse=lse)
    close(dq, qf.grad, dtype, "fused dq")
    close(dkv[..., :D], kvf.grad[..., :D], dtype, "fused dk")
    close(dkv[..., D:], kvf.grad[..., D:], dtype, "fused dv")
    # the materialised backward (no o / lse) agrees
    dq2, dkv2 = torch.empty_like(qd), torch.empty_like(kvd)
    ops.attention_bwd(qd, kvd[..., :D], kvd[..., D:], do.to(DEV, dtype), H, dq2, dkv2[..., :D], dkv2[..., D:])
    close(dq, dq2, dtype, "fused vs materialised dq")
    close(dkv, dkv2, dtype, "fused vs materialised dkv")


@pytest.mark.parametrize("B,Nq,Nkv,H", [(2, 70, 70, 2), (1, 1297, 1297, 4), (2, 333, 100, 4), (1, 64, 257, 3), (2, 4096, 256, 1), (1, 33, 65, 1)])
def test_attention_flash_forward_kernels_agree(B, Nq, Nkv, H):
    """The round-3 forward (32-query waves, four per SIMD) against the round-2 kernel (64-query waves): with the exact online
    softmax (defer 0) outputs and LSE are BIT-IDENTICAL (same MFMAs in the same order, same f32 arithmetic); with the deferred
    running maximum (the default: the maximum is only raised when a tile exceeds it by more than 2^6) they agree to bf16
    rounding, also when one late key dominates a row (forces the rescale branch after many deferred tiles)."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_set_flash_fwd.argtypes = [ctypes.c_int, ctypes.c_float]
    dtype, hd = torch.bfloat16, 64
    D = H * hd
    qh = rnd(B, Nq, D) * 1.2
    kvh = rnd(B, Nkv, 2 * D, seed=1) * 1.2
    qh[0, Nq // 2, :hd] = 4.0
    kvh[0, Nkv - 3, :hd] = 4.0            # a late key aligned with that query: score 128 after many small tiles
    qd, kvd = q(qh, dtype).to(DEV, dtype), q(kvh, dtype).to(DEV, dtype)
    outs = {}
    try:
        for tag, ver, defer in (("r2", 2, 0.0), ("r3 exact", 3, 0.0), ("r3 deferred", 3, 6.0), ("r5 exact", 4, 0.0),
                                ("r5 deferred", 4, 6.0), ("r5q exact", 5, 0.0), ("r5q deferred", 5, 6.0)):
            lib.gdl_debug_set_flash_fwd(ver, defer)
            outs[tag] = ops.attention_flash(qd, kvd[..., :D], kvd[..., D:], H, return_lse=True)
    finally:
        lib.gdl_debug_set_flash_fwd(-1, 6.0)      # back to the build's default forward
    assert torch.equal(outs["r3 exact"][0], outs["r2"][0]) and torch.equal(outs["r3 exact"][1], outs["r2"][1])
    # round 5 (S of the next key tile issued before the softmax of the current one; Q fragments in registers / in the LDS):
    # the same MFMAs and f32 operations in the same order per query row -- bit-identical, exact and deferred
    for r5 in ("r5", "r5q"):
        assert torch.equal(outs[f"{r5} exact"][0], outs["r2"][0]) and torch.equal(outs[f"{r5} exact"][1], outs["r2"][1]), r5
        assert torch.equal(outs[f"{r5} deferred"][0], outs["r3 deferred"][0]), r5
        assert torch.equal(outs[f"{r5} deferred"][1], outs["r3 deferred"][1]), r5

    def heads(t, n):
        return t.float().cpu().reshape(B, n, H, hd).transpose(1, 2)
    sc = heads(qd, Nq) @ heads(kvd[..., :D], Nkv).transpose(-1, -2) * hd ** -0.5
    ref = (sc.softmax(-1) @ heads(kvd[..., D:], Nkv)).transpose(1, 2).reshape(B, Nq, D)
    close(outs["r3 deferred"][0], ref, dtype, "deferred-maximum forward")
    close(outs["r3 deferred"][1], torch.logsumexp(sc, -1), torch.float32, "deferred-maximum lse", scale=max(1.0, sc.abs().max().item()) * 20)
    err2 = (outs["r2"][0].float().cpu() - ref).abs().max().item()
    err3 = (outs["r3 deferred"][0].float().cpu() - ref).abs().max().item()
    assert err3 <= 2.0 * err2 + 1e-3, (err2, err3)


def test_attention_flash_reference_drift_and_extremes():
    """Deferred running maximum (the default forward raises it only when a tile exceeds it by 2^6): (a) scores that creep up by a
    few units per key tile (many deferred tiles, then a raise); (b) later tiles far BELOW the maximum (their probabilities
    underflow, as in the exact softmax); (c) scores beyond +-300 (exp2 of a raw score would overflow f32) -- against an f64 softmax."""
    H, hd, N = 1, 64, 64 * 9 + 5
    base = rnd(1, N, 3 * hd) * 0.3
    cases = {}
    up = base.clone()
    up[0, :, hd:2 * hd] += torch.linspace(0, 6.0, N).view(N, 1) * 0.5      # key norm grows along the sequence
    up[0, :, :hd] += 0.5
    cases["creeping up"] = up
    down = base.clone()
    down[0, :64, hd:2 * hd] += 3.0                                          # the first tile holds the dominant keys
    down[0, :, :hd] += 1.0
    cases["first tile dominates"] = down
    big = base.clone() * 40.0
    cases["huge scores"] = big
    for name, qkv in cases.items():
        qkv = q(qkv, torch.bfloat16)
        o, lse = ops.attention_flash(*ops.split_qkv(qkv.to(DEV, torch.bfloat16)), H, return_lse=True)
        qq, kk, vv = (t.double() for t in qkv.view(1, N, 3, H, hd).permute(2, 0, 3, 1, 4))
        sc = qq @ kk.transpose(-1, -2) * hd ** -0.5
        ref = (sc.softmax(-1) @ vv).transpose(1, 2).reshape(1, N, hd)
        assert torch.isfinite(o.float()).all(), name
        close(o, ref.float(), torch.bfloat16, f"forward, {name}")
        close(lse, torch.logsumexp(sc, -1).float(), torch.float32, f"lse, {name}", scale=max(1.0, sc.abs().max().item()) * 20)


def test_attention_flash_spike():
    """one dominant key late in the sequence forces the online-softmax rescale branch."""
    B, N, H, hd = 1, 300, 1, 64
    qkv = rnd(B, N, 3 * hd) * 0.5
    qkv[0, 5, :hd] = 4.0            # query 5
    qkv[0, 260, hd:2 * hd] = 4.0    # key 260 aligned with it -> score 1024/8 = 128
    qkv = q(qkv, torch.bfloat16)
    o = ops.attention_flash(*ops.split_qkv(qkv.to(DEV, torch.bfloat16)), H)
    qq, kk, vv = qkv.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, N, hd)
    close(o, ref, torch.bfloat16, "attention_flash spike")


@pytest.mark.parametrize("out_dtype", DTYPES)
@pytest.mark.parametrize("D", [32, 64, 96, 128, 160, 768, 1024])
def test_layernorm(out_dtype, D):
    x, g, b = rnd(5, 33, D) * 3 + 1, rnd(D, seed=1), rnd(D, seed=2)
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, out_dtype)
    close(y, F.layer_norm(x, (D,), g, b, 1e-5), out_dtype, "layernorm")
    xs = rnd(4, 10, D).to(DEV)[:, 1:, :]   # strided rows
    y = ops.layernorm(xs, g.to(DEV), b.to(DEV), 1e-5, torch.float32)
    close(y, F.layer_norm(xs.cpu(), (D,), g, b, 1e-5), torch.float32, "layernorm strided")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 9, 7, 64), (2, 1, 1, 256), (3, 24, 24, 128), (2, 31, 9, 32), (1, 17, 23, 320),
                                     # >= 4096 pixels: the backward sums take the 16-byte-load kernel when C / 8 divides 256 or 192
                                     (4, 36, 36, 768), (2, 64, 64, 256), (2, 48, 48, 96), (1, 80, 80, 40), (1, 67, 67, 64)])
def test_batchnorm_train(dtype, B, H, W, C):
    x = q(rnd(B, H, W, C) * 2 + 0.5, dtype)
    g, b = rnd(C, seed=1), rnd(C, seed=2)
    rm, rv = rnd(C, seed=3) * 0.1, rnd(C, seed=4).abs() + 0.5
    dy = q(rnd(B, H, W, C, seed=5), dtype)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    yr = F.relu(F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5))
    yr.backward(dy.permute(0, 3, 1, 2))
    xd = x.to(DEV, dtype)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    mean, var = ops.bn_stats(xd, rmd, rvd, 0.1)
    close(rmd, rm_ref, torch.float32, "running_mean")
    close(rvd, rv_ref, torch.float32, "running_var")
    y = ops.bn_apply(xd, mean, var, g.to(DEV), b.to(DEV), 1e-5, True)
    close(y.permute(0, 3, 1, 2), yr, dtype, "bn fwd")
    dyd = dy.to(DEV, dtype)
    dg, db = ops.bn_bwd_reduce(xd, dyd, mean, var, g.to(DEV), b.to(DEV), 1e-5, True)
    if B * H * W > 2:  # with 2 samples xhat = +-1 and the reference itself is ill-conditioned
        close(dg, gr.grad, dtype, "dgamma")
        close(db, br.grad, dtype, "dbeta")
        dx = ops.bn_bwd_dx(xd, dyd, mean, var, g.to(DEV), b.to(DEV), 1e-5, True, dg, db, B * H * W)
        close(dx.permute(0, 3, 1, 2), xr.grad, dtype, "bn dx", scale=xr.grad.abs().max().item() + 1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("B,H,W,C", [(4, 36, 36, 256), (4, 1, 1, 256), (2, 3, 3, 768), (4, 18, 18, 768), (3, 37, 5, 12),
                                     (8, 32, 32, 64), (1, 45, 91, 16)])
def test_batchnorm_small_map_single_launch(dtype, B, H, W, C, relu):
    """Round 5: gdl_bn_small_fwd / gdl_bn_small_bwd -- the whole train-mode BatchNorm(+ReLU) of a small map in one launch per
    direction.  Outputs, saved statistics and running estimates equal the multi-launch kernels' (same per-element f32
    expressions; the sums are formed in another order: f32 tolerance), and both match torch's batch_norm autograd."""
    x = q(rnd(B, H, W, C) * 2 + 0.5, dtype)
    g, b = rnd(C, seed=1), rnd(C, seed=2)
    rm, rv = rnd(C, seed=3) * 0.1, rnd(C, seed=4).abs() + 0.5
    dy = q(rnd(B, H, W, C, seed=5), dtype)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    yr = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    yr = F.relu(yr) if relu else yr
    yr.backward(dy.permute(0, 3, 1, 2))
    xd, gd, bd, dyd = x.to(DEV, dtype), g.to(DEV), b.to(DEV), dy.to(DEV, dtype)
    assert ops.bn_small_fits(xd)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y, mean, var = ops.bn_small_fwd(xd, gd, bd, 1e-5, relu, rmd, rvd, 0.1)
    rm2, rv2 = rm.to(DEV), rv.to(DEV)
    mean2, var2 = ops.bn_stats(xd, rm2, rv2, 0.1)
    y2 = ops.bn_apply(xd, mean2, var2, gd, bd, 1e-5, relu)
    close(mean, mean2, torch.float32, "mean vs multi-launch")
    close(var, var2, torch.float32, "var vs multi-launch", scale=max(var2.abs().max().item(), 1e-3))
    close(rmd, rm_ref, torch.float32, "running_mean")
    close(rvd, rv_ref, torch.float32, "running_var")
    close(y, y2, dtype, "fwd vs multi-launch")
    close(y.permute(0, 3, 1, 2), yr, dtype, "fwd vs torch")
    if B * H * W > 2:
        dx, dg, db = ops.bn_small_bwd(xd, dyd, mean2, var2, gd, bd, 1e-5, relu)
        dg2, db2 = ops.bn_bwd_reduce(xd, dyd, mean2, var2, gd, bd, 1e-5, relu)
        dx2 = ops.bn_bwd_dx(xd, dyd, mean2, var2, gd, bd, 1e-5, relu, dg2, db2, B * H * W)
        close(dg, dg2, torch.float32, "dgamma vs multi-launch", scale=max(dg2.abs().max().item(), 1.0))
        close(db, db2, torch.float32, "dbeta vs multi-launch", scale=max(db2.abs().max().item(), 1.0))
        close(dx, dx2, dtype, "dx vs multi-launch", scale=dx2.float().abs().max().item() + 1e-3)
        # vs torch: with the ReLU, an element whose bn(x) is at f32 round-off level may sit on the other side of zero in torch's
        # evaluation, and one such element moves its channel's sums by |dy| and |dy * xhat| (seen: 0.3 on a sum of 124).  The
        # check against the multi-launch kernels above covers every channel (same per-element expressions); against torch the
        # channels that hold a pre-activation within 1e-4 of zero (f64 evaluation) are left out, the rest compare at full tolerance
        ok = torch.ones(C, dtype=torch.bool)
        if relu:
            pre = F.batch_norm(xr.detach().double(), None, None, g.double(), b.double(), True, 0.0, 1e-5)
            ok = (pre.abs() >= 1e-4).all(dim=3).all(dim=2).all(dim=0)
            assert int(ok.sum()) >= C // 2, "too few channels left to compare"
        close(dg.cpu()[ok], gr.grad[ok], dtype, "dgamma", scale=max(gr.grad.abs().max().item(), 1.0))
        close(db.cpu()[ok], br.grad[ok], dtype, "dbeta", scale=max(br.grad.abs().max().item(), 1.0))
        close(dx.cpu()[..., ok].permute(0, 3, 1, 2), xr.grad[:, ok], dtype, "dx vs torch", scale=xr.grad.abs().max().item() + 1e-3)
        # in place over x (what the ConvModule node does with its saved convolution output)
        xin = xd.clone()
        dx3, _, _ = ops.bn_small_bwd(xin, dyd, mean2, var2, gd, bd, 1e-5, relu, out=xin)
        assert dx3.data_ptr() == xin.data_ptr() and torch.equal(dx3, dx)


def test_bn_fold_matches_eval_bn():
    C = 96
    g, b, rm, rv = rnd(C), rnd(C, seed=1), rnd(C, seed=2), rnd(C, seed=3).abs() + 0.3
    x = rnd(2, C, 5, 5)
    scale, shift = ops.bn_fold(g.to(DEV), b.to(DEV), rm.to(DEV), rv.to(DEV), 1e-5)
    ref = F.batch_norm(x, rm, rv, g, b, False, 0.1, 1e-5)
    close(x * scale.cpu().view(1, C, 1, 1) + shift.cpu().view(1, C, 1, 1), ref, torch.float32, "bn_fold")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hi,ho", [(9, 36), (18, 36), (36, 18), (9, 32), (5, 5), (1, 6), (36, 128),
                                   # pyramid pooling branches (factors 18 / 9 / 6): the backward takes the window-parallel kernel
                                   (1, 18), (2, 18), (3, 18), (2, 37)])
def test_bilinear(dtype, hi, ho):
    B, C = 2, 64
    x = q(rnd(B, hi, hi, C), dtype)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = F.interpolate(xr, size=(ho, ho), mode="bilinear", align_corners=False)
    y = ops.bilinear(x.to(DEV, dtype), (ho, ho))
    close(y.permute(0, 3, 1, 2), ref, dtype, "bilinear fwd")
    dy = q(rnd(B, ho, ho, C, seed=3), dtype)
    ref.backward(dy.permute(0, 3, 1, 2))
    dx = ops.bilinear_bwd(dy.to(DEV, dtype), (hi, hi))
    close(dx.permute(0, 3, 1, 2), xr.grad, dtype, "bilinear bwd")
    # accumulate into a channel slice
    buf = torch.ones(B, ho, ho, 2 * C, device=DEV, dtype=dtype)
    ops.bilinear(x.to(DEV, dtype), (ho, ho), out=buf[..., C:], accumulate=True)
    close(buf[..., C:].permute(0, 3, 1, 2), ref + 1, dtype, "bilinear accumulate")
    assert (buf[..., :C] == 1).all()


@pytest.mark.parametrize("hi,wi,f,C", [(9, 13, 2, 64), (7, 5, 4, 72), (2, 2, 4, 8), (36, 36, 2, 256), (17, 3, 4, 16)])
def test_bilinear_integer_factor_gap_kernel(hi, wi, f, C):
    """bf16 upsampling by exactly 2 or 4 (FPN top-down path, the concat levels): a thread owns the F x F outputs between four
    input pixels (bilinear_fwd8_gap_kernel) -- bit-identical to the row kernel it replaces (same indices, weights, expression),
    plain and accumulating into a channel slice, and close to F.interpolate."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_set_flat_resample.argtypes = [ctypes.c_int]
    B, dtype = 3, torch.bfloat16
    x = q(rnd(B, hi, wi, C), dtype)
    size = (f * hi, f * wi)
    xd = x.to(DEV, dtype)
    base = q(rnd(B, size[0], size[1], 2 * C, seed=5), dtype).to(DEV, dtype)
    dy = q(rnd(B, size[0], size[1], C, seed=6), dtype)
    dyd = dy.to(DEV, dtype)
    outs = {}
    try:
        for mode in (0, 2):                                   # 0 = default (gap kernel), 2 = the row kernel
            lib.gdl_debug_set_flat_resample(mode)
            buf = base.clone()
            ops.bilinear(xd, size, out=buf[..., C:], accumulate=True)
            dx = ops.bilinear_bwd(dyd, (hi, wi))
            outs[mode] = (ops.bilinear(xd, size), buf, dx)
    finally:
        lib.gdl_debug_set_flat_resample(0)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    # base + resize(x) in one pass (gdl_bilinear_fwd_add, the FPN top-down add) == copy, then accumulate; base untouched
    half = base[..., :C].contiguous()
    keep = half.clone()
    want = half.clone()
    ops.bilinear(xd, size, out=want, accumulate=True)
    assert torch.equal(ops.bilinear_add(half, xd), want) and torch.equal(half, keep)
    # backward: a thread walks down a column of input pixels, every gradient row loaded once for the two input rows it feeds
    # (bilinear_bwd8_walk_kernel) -- bit-identical to the row kernel
    assert torch.equal(outs[0][2], outs[2][2])
    xr = x.permute(0, 3, 1, 2).float().clone().requires_grad_(True)
    F.interpolate(xr, size=size, mode="bilinear", align_corners=False).backward(dy.permute(0, 3, 1, 2).float())
    close(outs[0][2].permute(0, 3, 1, 2), xr.grad, dtype, "bilinear walk kernel (backward)")
    ref = F.interpolate(x.permute(0, 3, 1, 2).float(), size=size, mode="bilinear", align_corners=False)
    close(outs[0][0].permute(0, 3, 1, 2), ref, dtype, "bilinear gap kernel")
    assert torch.equal(outs[0][1][..., :C], base[..., :C])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hi,s", [(18, 1), (18, 2), (18, 3), (18, 6), (4, 6), (4, 3), (36, 6)])
def test_adaptive_avgpool(dtype, hi, s):
    B, C = 2, 64
    x = q(rnd(B, hi, hi, C), dtype)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xr, s)
    y = ops.adaptive_avgpool(x.to(DEV, dtype), s)
    close(y.permute(0, 3, 1, 2), ref, dtype, "avgpool fwd")
    dy = q(rnd(B, s, s, C, seed=2), dtype)
    ref.backward(dy.permute(0, 3, 1, 2))
    dx = ops.adaptive_avgpool_bwd(dy.to(DEV, dtype), (hi, hi))
    close(dx.permute(0, 3, 1, 2), xr.grad, dtype, "avgpool bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,N,R", [(2, 12, 12, 64, 64, 3), (2, 9, 11, 128, 256, 3), (4, 8, 8, 256, 128, 1),
                                         (2, 1, 1, 128, 256, 1), (1, 40, 40, 64, 64, 3),
                                         (2, 13, 11, 256, 512, 3), (3, 9, 9, 264, 320, 1), (1, 33, 35, 512, 256, 3)])
def test_conv_backward(dtype, B, H, W, C, N, R):
    x = q(rnd(B, C, H, W), dtype).requires_grad_(True)
    w = q(rnd(N, C, R, R, seed=1) * 0.1, dtype).requires_grad_(True)
    dy = q(rnd(B, N, H, W, seed=2), dtype)
    F.conv2d(x, w, padding=R // 2).backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    dw = ops.conv_wgrad(xn, dyn, R=R, S=R, pad=R // 2)
    close(dw.view(N, R, R, C).permute(0, 3, 1, 2), w.grad, dtype, "wgrad")
    wq = w.detach().permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV)
    wd = ops.pack_dgrad(wq, N, R * R, C, dtype)
    dx = ops.conv_gemm(dyn, wd, R=R, S=R, pad=R - 1 - R // 2, out_dtype=torch.float32)
    close(dx.permute(0, 3, 1, 2), x.grad, dtype, "dgrad")


@pytest.mark.parametrize("dtype,C,N", [(torch.float32, 64, 64), (torch.bfloat16, 256, 256)])
def test_wgrad_strided_dy(dtype, C, N):
    """dy as a channel slice of a wider gradient buffer (concat backward); the bf16 case runs the 256^2 kernel."""
    B, H, W = 2, 10, 10
    x = q(rnd(B, H, W, C), dtype)
    big = q(rnd(B, H, W, 3 * N, seed=1), dtype)
    dy = big.to(DEV, dtype)[..., N:2 * N]
    dw = ops.conv_wgrad(x.to(DEV, dtype), dy, R=3, S=3, pad=1)
    xr = x.permute(0, 3, 1, 2)
    w = torch.zeros(N, C, 3, 3, requires_grad=True)
    F.conv2d(xr, w, padding=1).backward(big[..., N:2 * N].permute(0, 3, 1, 2))
    close(dw.view(N, 3, 3, C).permute(0, 3, 1, 2), w.grad, dtype, "wgrad strided dy")


@pytest.mark.parametrize("B,H,W,C,N", [(2, 5, 64, 64, 64), (1, 3, 128, 128, 32), (2, 1, 64, 72, 64), (1, 24, 64, 192, 128),
                                       (3, 64, 64, 64, 16),
                                       # widths that are not multiples of 64 / 16: 144 -> 3 x 48, 72 -> 48 + 24, 100 -> 64 + 36
                                       (2, 7, 144, 256, 256), (2, 9, 72, 64, 128), (1, 11, 36, 320, 64), (2, 6, 100, 64, 64),
                                       (1, 5, 33, 64, 40), (4, 72, 144, 64, 64), (3, 37, 64, 128, 64)])
def test_wgrad_row_segment_kernel(B, H, W, C, N):
    """3x3 convs on maps at least 32 pixels wide take the row-segment kernel (all nine taps per block): checked
    against autograd and against the per-tap kernel it replaces (same products, different summation order)."""
    from gdlhip import _lib
    dtype = torch.bfloat16
    x = q(rnd(B, C, H, W), dtype)
    dy = q(rnd(B, N, H, W, seed=2), dtype)
    w = torch.zeros(N, C, 3, 3, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(dy)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    dw = ops.conv_wgrad(xn, dyn, R=3, S=3, pad=1)
    close(dw.view(N, 3, 3, C).permute(0, 3, 1, 2), w.grad, dtype, "row-segment wgrad")
    lib = _lib.load()
    lib.gdl_debug_force_wgrad_small(3)
    try:
        old = ops.conv_wgrad(xn, dyn, R=3, S=3, pad=1)
    finally:
        lib.gdl_debug_force_wgrad_small(0)
    assert (dw - old).abs().max().item() <= 1e-4 * old.abs().max().item()
    # round 5 (the default): all (n, c) tiles of one pixel range dealt to one XCD -- a relabelling of the workgroups, so the
    # launch-order form (hook 0) must give a bit-identical result
    import ctypes
    lib.gdl_debug_set_wgrad_rows_xcd.argtypes = [ctypes.c_int]
    lib.gdl_debug_set_wgrad_rows_xcd(0)
    try:
        in_launch_order = ops.conv_wgrad(xn, dyn, R=3, S=3, pad=1)
    finally:
        lib.gdl_debug_set_wgrad_rows_xcd(1)
    assert torch.equal(in_launch_order, dw)
    # accumulate into an existing gradient, dy as a channel slice of a wider buffer
    big = torch.zeros(B, H, W, N + 16, device=DEV, dtype=dtype)
    big[..., 8:8 + N] = dyn
    acc = torch.ones_like(dw)
    ops.conv_wgrad(xn, big[..., 8:8 + N], R=3, S=3, pad=1, dw=acc, accumulate=True)
    assert (acc - 1 - dw).abs().max().item() <= 1e-3 * dw.abs().max().item() + 1e-5


@pytest.mark.parametrize("B,K,hi,wi,ho,wo", [(2, 5, 9, 9, 32, 32), (2, 5, 36, 36, 128, 128), (1, 3, 7, 5, 7, 5), (3, 16, 6, 10, 50, 41),
                                             (2, 2, 4, 4, 64, 64), (1, 5, 144, 144, 512, 512), (2, 5, 16, 16, 512, 512), (2, 5, 18, 18, 512, 512), (1, 12, 3, 3, 96, 160)])
def test_dice_loss_from_low_resolution_logits(B, K, hi, wi, ho, wo):
    """Round 5: gdl_dice_loss_lowres_fwd / _bwd -- DiceLoss(F.interpolate(low, size, bilinear)) and its gradient w.r.t. ``low``
    without the [B, K, H, W] logits (dofa.py:89-105 + segmentation_dofa.py:226-229).  Against (a) the materialised path it replaces
    (gdl_upsample_logits -> gdl_dice_loss_fwd / _bwd -> gdl_upsample_logits_bwd: same per-pixel expressions) and (b) torch autograd
    of interpolate -> softmax -> the smp Dice formula in f64; with an upstream scale (the 0.4 of the auxiliary head) and a class
    that does not occur in the target."""
    from oracle.model import dice_loss_multiclass
    low = rnd(B, hi, wi, K, seed=3) * 2.0
    tgt = torch.randint(0, max(K - 1, 1), (B, ho, wo), generator=torch.Generator().manual_seed(4))     # class K-1 never occurs
    lowd, tgtd = low.to(DEV), tgt.to(DEV)
    assert ops.dice_lowres_ok(lowd, (ho, wo))
    loss, sums = ops.dice_loss_lowres_fwd(lowd, tgtd, (ho, wo))
    full = ops.upsample_logits(lowd, (ho, wo))
    loss2, sums2 = ops.dice_loss_fwd(full, tgtd)
    assert abs(loss.item() - loss2.item()) <= 1e-6 and torch.allclose(sums, sums2, rtol=2e-6, atol=1e-3)
    up = torch.tensor(0.4, device=DEV)
    dlow = ops.dice_loss_lowres_bwd(lowd, tgtd, (ho, wo), sums, up)
    dlow2 = ops.upsample_logits_bwd(ops.dice_loss_bwd(full, tgtd, sums2, up), (hi, wi))
    close(dlow, dlow2, torch.float32, "d low vs the materialised path", scale=dlow2.abs().max().item() + 1e-12)
    # the definition
    lr = low.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = dice_loss_multiclass(F.interpolate(lr, size=(ho, wo), mode="bilinear", align_corners=False), tgt)
    (0.4 * ref).backward()
    assert abs(loss.item() - ref.item()) <= 2e-6, (loss.item(), ref.item())
    close(dlow.permute(0, 3, 1, 2), lr.grad.float(), torch.float32, "d low vs torch", scale=lr.grad.abs().max().item() + 1e-12)
    # through the module: LowresLogits -> DiceLoss == DiceLoss(materialised logits), gradients into the low-resolution map
    crit = gnn.DiceLoss(mode="multiclass")
    a = lowd.clone().requires_grad_(True)
    b_ = lowd.clone().requires_grad_(True)
    la = crit(gnn.LowresLogits(a, (ho, wo)), tgtd)
    lb = crit(gnn.LowresLogits(b_, (ho, wo)).materialise(), tgtd)
    (0.4 * la).backward()
    (0.4 * lb).backward()
    assert abs(la.item() - lb.item()) <= 1e-6
    close(a.grad, b_.grad, torch.float32, "module path", scale=b_.grad.abs().max().item() + 1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [5, 1, 12, 16])
def test_head_and_logit_upsample(dtype, K):
    B, H, W, C, HO = 2, 9, 9, 256, 32
    feat = q(rnd(B, H, W, C), dtype)
    w, bias = rnd(K, C, seed=1) * 0.1, rnd(K, seed=2)
    cs = (torch.rand(B, C, generator=torch.Generator().manual_seed(3)) < 0.9).float() / 0.9
    fr = feat.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    low = F.conv2d(fr * cs[:, :, None, None], wr.view(K, C, 1, 1), br)
    ref = F.interpolate(low, size=(HO, HO), mode="bilinear", align_corners=False)
    got_low = ops.head_1x1(feat.to(DEV, dtype), w.to(DEV), bias.to(DEV), cs.to(DEV))
    close(got_low.permute(0, 3, 1, 2), low, torch.float32 if dtype == torch.float32 else dtype, "head 1x1")
    got = ops.upsample_logits(got_low, (HO, HO))
    close(got, ref, dtype, "upsample logits")
    g = rnd(B, K, HO, HO, seed=4)
    ref.backward(g)
    dlow = ops.upsample_logits_bwd(g.to(DEV), (H, W))
    dfeat, dw, db = ops.head_1x1_bwd(feat.to(DEV, dtype), dlow, w.to(DEV), cs.to(DEV))
    close(dfeat.permute(0, 3, 1, 2), fr.grad, dtype, "head dfeat")
    close(dw, wr.grad, dtype, "head dw")
    close(db, br.grad, dtype, "head db")


@pytest.mark.parametrize("scale", [False, True])
@pytest.mark.parametrize("C", [256, 128, 768, 512, 1024])
@pytest.mark.parametrize("K,B,H,W", [(5, 2, 36, 36), (1, 1, 35, 31), (12, 1, 40, 40), (16, 3, 21, 17), (5, 4, 144, 144)])
def test_head_mfma_and_register_weight_kernels(K, B, H, W, C, scale):
    """Round 5: the classifier head as a skinny MFMA GEMM (gdl_head_1x1 for dense bf16 features: 16-pixel tiles staged through a
    wave-private LDS slot, f32 weights as bf16 hi + lo fragments; C = 128 / 256 one wave per tile, C = 512 / 768 / 1024 -- SegFormer's
    decoder, segformer_mlp.py:64-65 -- 256-channel slices on the waves of a workgroup; a Dropout2d scale folded into the weights per
    image) and its gradients with register-resident weights (dofa.py:89-96 / segmentation_head.py / fcn_head.py:73).  Against the
    wave-per-pixel kernels they replace (forward: f32-grade agreement; feature gradient: to a bf16 ulp; weight gradient from up to 2048
    partial rows) and against torch in f64; ragged last tile (P % 16 != 0), images that are not a multiple of 16 pixels (scale: old path)."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_set_head_mfma.argtypes = [ctypes.c_int]
    if B * H * W * C > 2 ** 26 and C > 768:
        pytest.skip("largest map only up to 768 channels")
    feat = q(rnd(B, H, W, C), torch.bfloat16)
    w, bias = rnd(K, C, seed=1) * 0.1, rnd(K, seed=2)
    cs = (torch.rand(B, C, generator=torch.Generator().manual_seed(3)) < 0.9).float() / 0.9 if scale else None
    csd = None if cs is None else cs.to(DEV)
    fd, wd, bd = feat.to(DEV, torch.bfloat16), w.to(DEV), bias.to(DEV)
    dlow = (rnd(B, H, W, K, seed=5) * 0.01).to(DEV)
    try:
        lib.gdl_debug_set_head_mfma(0)
        low_old = ops.head_1x1(fd, wd, bd, csd)
        dfeat_old, dw_old, db_old = ops.head_1x1_bwd(fd, dlow, wd, csd)
    finally:
        lib.gdl_debug_set_head_mfma(1)
    low = ops.head_1x1(fd, wd, bd, csd)
    dfeat, dw, db = ops.head_1x1_bwd(fd, dlow, wd, csd)
    fs = feat.double() if cs is None else feat.double() * cs.double()[:, None, None, :]
    ref = torch.einsum("bhwc,kc->bhwk", fs, w.double()) + bias.double()
    sc = ref.abs().max().item()
    assert (low.cpu().double() - ref).abs().max().item() <= 2e-5 * sc, "MFMA head vs f64"
    assert (low - low_old).abs().max().item() <= 2e-5 * sc, "MFMA head vs the wave-per-pixel kernel"
    dfeat_ref = torch.einsum("bhwk,kc->bhwc", dlow.cpu().double(), w.double())
    if cs is not None:
        dfeat_ref = dfeat_ref * cs.double()[:, None, None, :]
    close(dfeat, dfeat_ref.float(), torch.bfloat16, "head dfeat vs f64", scale=dfeat_ref.abs().max().item())
    # (same products in the same order; the two kernels may differ in which multiply-adds the compiler fused: one bf16 ulp)
    assert (dfeat.float() - dfeat_old.float()).abs().max().item() <= 2.0 ** -7 * dfeat_ref.abs().max().item()
    close(dw, dw_old, torch.float32, "head dw", scale=dw_old.abs().max().item())
    close(db, db_old, torch.float32, "head db", scale=db_old.abs().max().item())
    dw_ref = torch.einsum("bhwk,bhwc->kc", dlow.cpu().double(), fs)
    close(dw, dw_ref.float(), torch.float32, "head dw vs f64", scale=dw_ref.abs().max().item())
    close(db, dlow.cpu().double().sum((0, 1, 2)).float(), torch.float32, "head db vs f64", scale=max(db_old.abs().max().item(), 1e-3))
    # no bias
    low_nb = ops.head_1x1(fd, wd, None, csd)
    assert (low_nb.cpu().double() + bias.double() - ref).abs().max().item() <= 2e-5 * sc


@pytest.mark.parametrize("K", [2, 9, 16])
def test_classifier_tail_many_classes(K):
    """Dice loss, softmax->argmax and class probabilities beyond 8 classes (land-cover legends): up to 16."""
    import oracle
    B, H = 2, 40
    logits = (rnd(B, K, H, H) * 2).requires_grad_(True)
    y = torch.randint(0, K, (B, H, H), generator=torch.Generator().manual_seed(K))
    ref = oracle.model.dice_loss_multiclass(logits, y)
    ref.backward()
    ld = logits.detach().to(DEV).requires_grad_(True)
    loss = gnn.DiceLoss()(ld, y.to(DEV))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    close(ld.grad, logits.grad, torch.float32, "dice grad")
    assert torch.equal(ops.softmax_argmax(ld.detach()).cpu(), logits.detach().softmax(1).argmax(1))
    close(ops.class_probs(ld.detach()), logits.detach().softmax(1), torch.float32, "class probs")
    from gdlhip._lib import GdlHipError
    with pytest.raises(GdlHipError, match="1..16"):            # a loud, typed failure -- never a silent fallback
        ops.softmax_argmax(torch.zeros(1, 17, 8, 8, device=DEV))


def test_softmax_argmax_bit_exact():
    logits = rnd(2, 5, 64, 64) * 4
    logits[0, 1, :8] = logits[0, 3, :8]          # exact ties -> first index must win
    got = ops.softmax_argmax(logits.to(DEV)).cpu()
    ref = logits.softmax(dim=1).argmax(dim=1)
    assert got.dtype == torch.int64 and torch.equal(got, ref)


def test_dice_loss():
    import oracle
    B, K, H = 2, 5, 48
    logits = (rnd(B, K, H, H) * 2).requires_grad_(True)
    y = torch.randint(0, 4, (B, H, H), generator=torch.Generator().manual_seed(1))  # class 4 absent
    ref = oracle.model.dice_loss_multiclass(logits, y)
    (ref * 0.4).backward()
    ld = logits.detach().to(DEV).requires_grad_(True)
    loss = gnn.DiceLoss()(ld, y.to(DEV))
    (loss * 0.4).backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    close(ld.grad, logits.grad, torch.float32, "dice grad")


@pytest.mark.parametrize("empty", [False, True])
def test_dice_loss_binary(empty):
    """DiceLoss(mode="binary") of the reference's UNet++ config (num_classes 1), un-squeezed [B,1,H,W] mask."""
    import oracle
    B, H = 3, 40
    logits = (rnd(B, 1, H, H) * 3).requires_grad_(True)
    y = torch.zeros(B, 1, H, H, dtype=torch.int64) if empty else \
        torch.randint(0, 2, (B, 1, H, H), generator=torch.Generator().manual_seed(2))
    ref = oracle.model.dice_loss_binary(logits, y)
    (ref * 0.7).backward()
    ld = logits.detach().to(DEV).requires_grad_(True)
    loss = gnn.DiceLoss(mode="binary")(ld, y.to(DEV))
    (loss * 0.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    close(ld.grad, logits.grad, torch.float32, "binary dice grad")


def test_dice_loss_multiclass_accepts_unsqueezed_mask():
    logits = (rnd(2, 5, 24, 24) * 2).to(DEV)
    y = torch.randint(0, 5, (2, 1, 24, 24), generator=torch.Generator().manual_seed(3)).to(DEV)
    assert gnn.DiceLoss()(logits, y).item() == gnn.DiceLoss()(logits, y[:, 0]).item()


@pytest.mark.parametrize("B,h,w,H,W,K", [(2, 36, 36, 128, 128, 5), (3, 16, 16, 512, 512, 5), (2, 9, 13, 33, 40, 2), (1, 18, 18, 64, 64, 16),
                                        (2, 7, 7, 7, 7, 3)])
def test_upsample_argmax_bit_exact(B, h, w, H, W, K):
    """The validation / inference mask straight from the head's low-resolution map (gdl_upsample_argmax) == upsample_logits followed by
    softmax_argmax, in every pixel -- including maps with many near-ties (logits quantised to a coarse grid) -- and gnn.predict_mask
    takes that path for LowresLogits."""
    g = torch.Generator().manual_seed(B * 1000 + h * 10 + K)
    for scale, grid in ((2.0, None), (1.0, 0.25)):
        low = torch.randn(B, h, w, K, generator=g) * scale
        if grid:
            low = (low / grid).round() * grid
        low = low.to(DEV)
        want = ops.softmax_argmax(ops.upsample_logits(low, (H, W)))
        got = ops.upsample_argmax(low, (H, W))
        assert got.dtype == torch.int64 and torch.equal(got, want)
        assert torch.equal(gnn.predict_mask(gnn.LowresLogits(low, (H, W))), want)


def test_adam_and_clip():
    torch.manual_seed(0)
    ps = [torch.randn(1000, 3), torch.randn(64, 32, 3, 3).contiguous(memory_format=torch.channels_last)]
    gs = [torch.randn_like(p) * 3 for p in ps]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=6e-5)
    o_mine = gnn.FusedAdam(mine, lr=6e-5, max_grad_norm=1.0)
    for step in range(3):
        for r, m, g in zip(ref, mine, gs):
            r.grad = g.clone() * (step + 1)
            m.grad = (g.clone() * (step + 1)).to(DEV)
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        o_ref.step()
        o_mine.step()
    for r, m in zip(ref, mine):
        assert (m.detach().cpu() - r.detach()).abs().max().item() < 2e-6
    # gradients rewritten IN PLACE (DDP bucket views, zero_grad(set_to_none=False)): the chunk table is reused, not rebuilt;
    # a learning rate written by a scheduler still takes effect (it is a kernel argument, not part of the table)
    builds = o_mine.table_builds
    for step in range(3):
        for group in (o_ref.param_groups[0], o_mine.param_groups[0]):
            group["lr"] = 6e-5 * (step + 2)
        for r, m, g in zip(ref, mine, gs):
            r.grad = g.clone() * 0.1 * (step + 1)
            m.grad.copy_((g * 0.1 * (step + 1)).to(DEV))
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        o_ref.step()
        o_mine.step()
    assert o_mine.table_builds == builds
    for r, m in zip(ref, mine):
        assert (m.detach().cpu() - r.detach()).abs().max().item() < 4e-6


def test_adam_rewrites_the_bf16_gemm_operands(monkeypatch):
    """The optimizer's update kernel also rewrites the bf16 GEMM operand of each parameter (gnn.gemm_weight's cache entry), so
    the next forward finds it current: bit-identical to a fresh cast of the new parameter, no cast launch, and a parameter
    whose operand is NOT a plain cast (stored NCHW: the operand is a repack) keeps the rebuild-on-use path."""
    torch.manual_seed(1)
    lin = torch.randn(96, 200, device=DEV).requires_grad_(True)                                        # nn.Linear weight
    cl = torch.randn(64, 32, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    nchw = torch.randn(64, 32, 3, 3, device=DEV).requires_grad_(True)
    params = [lin, cl, nchw]
    opt = gnn.FusedAdam(params, lr=1e-2)
    first = [gnn.gemm_weight(p, torch.bfloat16) for p in params]      # what a forward under bf16 autocast creates
    casts = []
    real_cast = ops.cast
    monkeypatch.setattr(ops, "cast", lambda *a, **k: (casts.append(1), real_cast(*a, **k))[1])
    for p in params:
        p.grad = torch.zeros_like(p)
    for step in range(3):
        for p in params:
            p.grad.copy_(torch.randn_like(p))
        opt.step()
        casts.clear()
        now = [gnn.gemm_weight(p, torch.bfloat16) for p in params]
        assert len(casts) == 1, casts                                  # only the NCHW-stored parameter is re-cast
        assert now[0] is first[0] and now[1] is first[1] and now[2] is not first[2]
        assert torch.equal(now[0], lin.detach().to(torch.bfloat16))
        assert torch.equal(now[1], cl.detach().permute(0, 2, 3, 1).reshape(64, -1).to(torch.bfloat16))
        assert torch.equal(now[2], nchw.detach().permute(0, 2, 3, 1).reshape(64, -1).to(torch.bfloat16))
    assert opt.table_builds == 1
    # a parameter rewritten by someone else (load_state_dict, manual copy_) invalidates the operand as before
    with torch.no_grad():
        lin.mul_(2.0)
    assert torch.equal(gnn.gemm_weight(lin, torch.bfloat16), lin.detach().to(torch.bfloat16))


def test_adam_rebuilds_the_derived_conv_operands_in_one_launch(monkeypatch):
    """Behind its update the optimizer rebuilds every bf16 operand DERIVED from a conv parameter in another element order --
    channel slices and tap-major forms of the 3x3 parameters (UperNet's fpn_bottleneck over the concat, the neck's resized
    convolutions), data-gradient operands -- in ONE launch (gdl_multi_repack), in place, and marks the cache entries current:
    the next forward launches no repack / cast, and the operands are bit-identical to a rebuild by the separate launches."""
    torch.manual_seed(2)
    bf = torch.bfloat16
    w3 = torch.randn(72, 160, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)   # ragged tiles
    w1 = torch.randn(96, 40, 1, 1, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    nchw = torch.randn(64, 32, 3, 3, device=DEV).requires_grad_(True)            # not dense [N][T][C]: keeps the rebuild-on-use path
    frozen = torch.randn(64, 32, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last)
    params = [w3, w1, nchw]

    def forward_operands():
        return [gnn.tap_weight(w3, bf), gnn.tap_weight(w3, bf, 32, 96), gnn.slice_weight(w3, bf, 0, 32), gnn.slice_weight(w3, bf, 96, 160),
                gnn.dgrad_weight(w3, bf), gnn.dgrad_weight(w1, bf), gnn.dgrad_weight(nchw, bf), gnn.tap_weight(frozen, bf)]

    def reference():
        m3 = w3.detach().permute(0, 2, 3, 1).reshape(72, 9, 160)
        tap = lambda m, a, b: m[:, :, a:b].permute(1, 0, 2).reshape(9 * m.shape[0], b - a).to(bf)                     # noqa: E731
        sl = lambda m, a, b: m[:, :, a:b].reshape(m.shape[0], -1).to(bf)                                              # noqa: E731
        dg = lambda w: w.detach().permute(1, 2, 3, 0).flip(1, 2).reshape(w.shape[1], -1).to(bf)                       # noqa: E731
        return [tap(m3, 0, 160), tap(m3, 32, 96), sl(m3, 0, 32), sl(m3, 96, 160), dg(w3), dg(w1), dg(nchw),
                tap(frozen.permute(0, 2, 3, 1).reshape(64, 9, 32), 0, 32)]

    first = forward_operands()
    for got, want in zip(first, reference()):
        assert torch.equal(got, want)
    opt = gnn.FusedAdam(params, lr=1e-2)
    launches = []
    for name in ("cast", "pack_dgrad"):
        real = getattr(ops, name)
        monkeypatch.setattr(ops, name, lambda *a, _real=real, _n=name, **k: (launches.append(_n), _real(*a, **k))[1])
    for p in params:
        p.grad = torch.zeros_like(p)
    for step in range(3):
        for p in params:
            p.grad.copy_(torch.randn_like(p))
        opt.step()
        launches.clear()
        now = forward_operands()
        assert launches == ["pack_dgrad"], launches                    # only the NCHW-stored parameter is repacked on use
        assert all(a is b for a, b in zip(now[:6], first[:6])) and now[6] is not first[6] and now[7] is first[7]
        for i, (got, want) in enumerate(zip(now, reference())):
            assert torch.equal(got, want), (step, i)
    assert opt._repack is not None and opt._repack[1].shape[0] == 6
    # a parameter rewritten by someone else invalidates its operands as before; the optimizer then adopts the rebuilt tensors
    with torch.no_grad():
        w3.mul_(0.5)
    again = forward_operands()
    assert again[0] is not first[0]
    for got, want in zip(again, reference()):
        assert torch.equal(got, want)
    for p in params:
        p.grad.copy_(torch.randn_like(p))
    opt.step()
    for got, want in zip(forward_operands(), reference()):
        assert torch.equal(got, want)


def test_drop_path_scales_one_launch_per_pass():
    """gnn.drop_path_scales: all DropPath draws of an encoder pass at once -- rows are mask / keep with the block's own keep
    probability (timm drop_path, scale_by_keep), None for blocks that never drop."""
    torch.manual_seed(0)
    probs = [0.0, 0.05, 0.5]
    out = gnn.drop_path_scales(probs, 20000, torch.device(DEV))
    assert out[0] == (None, None)
    for p_, pair in zip(probs[1:], out[1:]):
        for t in pair:
            assert t.shape == (20000,) and t.is_contiguous() and t.dtype == torch.float32
            vals = t.unique().cpu()
            assert torch.allclose(vals, torch.tensor([0.0, 1.0 / (1.0 - p_)]))
            assert abs((t > 0).float().mean().item() - (1.0 - p_)) < 0.02
    assert not torch.equal(out[2][0], out[2][1])


def test_patch_embed_pieces():
    B, C, H, P, D = 2, 3, 56, 14, 64
    img = rnd(B, C, H, H)
    g = rnd(C, P * P * D, seed=1)
    w_ref = (g.view(C, P, P, D).permute(3, 0, 1, 2) * 0.01)
    ref = F.conv2d(img, w_ref, stride=P, padding=1).flatten(2).transpose(1, 2)
    gh = (H + 2 - P) // P + 1
    kpad = (C * P * P + 31) // 32 * 32
    cols = ops.patchify(img.to(DEV), P, 1, gh, gh, kpad, torch.float32)
    wq = ops.dofa_pack_kernel(g.to(DEV), C, P * P, D, 0.01, kpad, torch.float32)
    y = ops.linear(cols, wq).view(B, gh * gh, D)
    close(y, ref, torch.float32, "patch embed")


def test_normalize_u8_golden(golden_dir):
    import numpy as np
    g = np.load(golden_dir / "tensors_preprocess.npz")
    out = ops.normalize_u8(torch.from_numpy(g["u8"]).to(DEV), torch.from_numpy(g["mean"]).flatten().to(DEV),
                           torch.from_numpy(g["std"]).flatten().to(DEV))
    assert (out.cpu() - torch.from_numpy(g["out"])).abs().max().item() < 1e-6


def test_sincos_embed():
    from oracle.encoder import position_embedding
    wv = torch.tensor([0.665, 0.549, 0.481, 2.19])
    got = ops.sincos_embed((wv * 1000).to(DEV), 128)
    close(got, position_embedding(128, wv * 1000), torch.float32, "sincos", scale=1.0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gdlhip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.GdlHipError):
        ops.cast(torch.zeros(4, device=DEV), torch.bfloat16)
    assert not math.isnan(0.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 256), (1, 9, 7, 64), (2, 5, 70, 128), (1, 1, 1, 64), (2, 3, 33, 320)])
def test_dwconv3x3_gelu(dtype, B, H, W, C):
    x = q(rnd(B, C, H, W), dtype)
    w, b = rnd(C, 1, 3, 3, seed=1) * 0.3, rnd(C, seed=2)
    ref = F.gelu(F.conv2d(x, w, b, padding=1, groups=C))
    w9 = w.reshape(C, 9).t().contiguous().to(DEV)
    y = ops.dwconv3x3(x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype), w9, b.to(DEV), True)
    close(y.permute(0, 3, 1, 2), ref, dtype, "dwconv+gelu")


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_separate_kv(dtype):
    """SegFormer SR attention: Nq != Nkv, K/V packed in one [B,Nkv,2C] tensor."""
    B, Nq, Nkv, H, hd = 2, 200, 64, 2, 64
    C = H * hd
    qq, kv = q(rnd(B, Nq, C), dtype), q(rnd(B, Nkv, 2 * C, seed=1), dtype)
    qd, kvd = qq.to(DEV, dtype), kv.to(DEV, dtype)
    o = ops.attention(qd, kvd[..., :C], kvd[..., C:], H)
    k, v = kv[..., :C], kv[..., C:]
    ref = F.scaled_dot_product_attention(qq.view(B, Nq, H, hd).transpose(1, 2), k.reshape(B, Nkv, H, hd).transpose(1, 2),
                                         v.reshape(B, Nkv, H, hd).transpose(1, 2)).transpose(1, 2).reshape(B, Nq, C)
    close(o, ref, dtype, "attention q/kv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,s,p,C,N", [(3, 2, 1, 64, 128), (8, 8, 0, 64, 64), (2, 2, 0, 320, 320)])
def test_conv_strided(dtype, k, s, p, C, N):
    """MiT patch-embed (3x3/s2) and spatial-reduction (k = stride) convs."""
    B, H = 2, 16
    x, w = q(rnd(B, C, H, H), dtype), q(rnd(N, C, k, k, seed=1) * 0.05, dtype)
    ref = F.conv2d(x, w, stride=s, padding=p)
    y = ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype),
                      w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV, dtype), R=k, S=k, stride=s, pad=p,
                      out_dtype=torch.float32)
    close(y.permute(0, 3, 1, 2), ref, dtype, "strided conv")


def test_patchify_strided_stem():
    B, C, H, P, S, N = 2, 3, 32, 7, 4, 64
    img, w = rnd(B, C, H, H), rnd(N, C, P, P, seed=1) * 0.1
    ref = F.conv2d(img, w, stride=S, padding=3)
    g = (H + 6 - P) // S + 1
    kpad = (C * P * P + 31) // 32 * 32
    cols = ops.patchify(img.to(DEV), P, 3, g, g, kpad, torch.float32, stride=S)
    wq = torch.zeros(N, kpad)
    wq[:, : C * P * P] = w.reshape(N, -1)
    y = ops.linear(cols, wq.to(DEV)).view(B, g, g, N)
    close(y.permute(0, 3, 1, 2), ref, torch.float32, "7x7/s4 stem")


def test_iou_counts_bit_exact_and_mean_iou():
    """Integer work: the per-sample / per-class counts are bit-exact vs numpy; MeanIoU follows the restated
    torchmetrics formula (per-sample I/U, 0 where the union is empty, batch mean, mean over updates)."""
    import numpy as np
    from gdlhip.metrics import ClasswiseWrapper, MeanIoU
    g = torch.Generator().manual_seed(3)
    B, K, H, W = 3, 5, 96, 70
    pred = torch.randint(0, K, (B, H, W), generator=g)
    tgt = torch.randint(0, K, (B, H, W), generator=g)
    tgt[1][tgt[1] == 3] = 0                         # class 3 absent from sample 1's target
    pred[2] = 4                                     # a constant prediction
    counts = ops.iou_counts(pred.to(DEV), tgt.to(DEV), K).cpu().numpy()
    pn, tn = pred.numpy(), tgt.numpy()
    for b in range(B):
        for k in range(K):
            assert counts[b, 0, k] == np.sum((pn[b] == k) & (tn[b] == k))
            assert counts[b, 1, k] == np.sum(pn[b] == k)
            assert counts[b, 2, k] == np.sum(tn[b] == k)
    inter = counts[:, 0].astype(np.float64)
    union = (counts[:, 1] + counts[:, 2] - counts[:, 0]).astype(np.float64)
    ref = np.where(union > 0, inter / np.maximum(union, 1), 0.0).mean(0)
    m = ClasswiseWrapper(MeanIoU(num_classes=K, per_class=True, input_format="index"), labels=list("abcde")).to(DEV)
    out = m(pred.to(DEV), tgt.to(DEV))
    assert list(out) == [f"meaniou_{c}" for c in "abcde"]
    got = np.array([v.item() for v in out.values()])
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7)
    m.update(pred.to(DEV), tgt.to(DEV))
    np.testing.assert_allclose(np.array([v.item() for v in m.compute().values()]), ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,N", [(3, 19, 23, 128, 256), (2, 16, 16, 64, 512), (1, 40, 7, 192, 256)])
def test_conv3x3_shared_staging_kernel(dtype, B, H, W, C, N):
    """The 3x3 kernel that stages one activation tile for the three taps of a filter row (image-edge taps are
    zeroed per lane): forced on small multi-image shapes with M tails, vs F.conv2d and vs the generic kernel."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, 3, 3, seed=1) * 0.1, dtype)
    bias, resid = rnd(N, seed=2), q(rnd(B, N, H, W, seed=3), dtype)
    ref = F.relu(F.conv2d(x, w, bias, padding=1)) + resid
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wq = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV, dtype)
    rn = resid.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    outs = {}
    try:
        for v in (4, 1):
            lib.gdl_debug_force_conv_variant(v)
            outs[v] = ops.conv_gemm(xn, wq, R=3, S=3, pad=1, bias=bias.to(DEV), act=ops.ACT_RELU, resid=rn,
                                    out_dtype=torch.float32)
    finally:
        lib.gdl_debug_force_conv_variant(-1)
    close(outs[4].permute(0, 3, 1, 2), ref, dtype, "3x3 shared staging")
    assert (outs[4] - outs[1]).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("B,H,W,C,N,R,stride", [(2, 37, 23, 64, 128, 3, 1), (1, 50, 50, 768, 256, 1, 1), (3, 16, 16, 192, 200, 3, 1),
                                                (2, 21, 21, 128, 384, 3, 2), (1, 5, 5, 64, 64, 1, 1)])
@pytest.mark.parametrize("variant", [6, 8])
def test_conv_dual_resident_tile(B, H, W, C, N, R, stride, variant):
    """The 256 x 128 / four-wave tile of which two workgroups share a CU (variant 6, conv_gemm_dual.hip: weights single-buffered,
    two barriers per K-step) and the 256 x 256 tile with one wave per SIMD (variant 8, conv_gemm_w4.hip: every fragment read and
    DMA piece behind an MFMA, a tile's pieces spread over two K-steps), forced on small shapes: 1x1 dense and 3x3 / strided addressing, M and N tails, epilogue with bias +
    ReLU + residual -- vs F.conv2d and vs the 128^2 tile (same K order: identical sums)."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    dtype = torch.bfloat16
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, R, R, seed=1) * 0.1, dtype)
    Ho, Wo = (H + 2 * (R // 2) - R) // stride + 1, (W + 2 * (R // 2) - R) // stride + 1
    bias, resid = rnd(N, seed=2), q(rnd(B, N, Ho, Wo, seed=3), dtype)
    ref = F.relu(F.conv2d(x, w, bias, padding=R // 2, stride=stride)) + resid
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wq = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV, dtype)
    rn = resid.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    outs = {}
    try:
        for v in (variant, 1):
            lib.gdl_debug_force_conv_variant(v)
            for odt in (torch.float32, torch.bfloat16):
                outs[v, odt] = ops.conv_gemm(xn, wq, R=R, S=R, pad=R // 2, stride=stride, bias=bias.to(DEV), act=ops.ACT_RELU,
                                             resid=rn if odt == torch.bfloat16 else rn.float(), out_dtype=odt)
    finally:
        lib.gdl_debug_force_conv_variant(-1)
    close(outs[variant, torch.float32].permute(0, 3, 1, 2), ref, dtype, f"tile variant {variant}")
    for odt in (torch.float32, torch.bfloat16):
        assert torch.equal(outs[variant, odt], outs[1, odt]), f"variant {variant} vs 128^2 tile differ ({odt})"


@pytest.mark.parametrize("B,H,W,C,N,R,K_steps", [(4, 18, 18, 1792, 256, 3, 252),    # UperNet's pyramid-pooling bottleneck at batch 4: 84 tiles
                                                (4, 18, 18, 768, 768, 3, 108),     # the neck's 3x3 at 18^2: 252 tiles
                                                (1, 13, 9, 704, 72, 3, 99),        # M and N tails, 11 chunks per tap
                                                (2, 1, 70, 6144, 200, 1, 96)])     # dense 1x1 rows
def test_conv_four_stage_tile(B, H, W, C, N, R, K_steps):
    """The 64^2 tile with FOUR LDS stages (variant 12: the DMA of tile t+4 is issued at K-step t): what the planner picks for
    bf16 layers with at most one round of tiles and >= 96 K-steps -- vs F.conv2d, and bit-identical to the two-stage 64^2 tile
    (same tiles, same K order), with bf16 and f32 outputs, bias + ReLU + residual."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    dtype = torch.bfloat16
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, R, R, seed=1) * 0.05, dtype)
    bias, resid = rnd(N, seed=2), q(rnd(B, N, H, W, seed=3), dtype)
    ref = F.relu(F.conv2d(x, w, bias, padding=R // 2)) + resid
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wq = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV, dtype)
    rn = resid.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    assert R * R * C // 64 == K_steps
    outs = {}
    ops.TIMER = timer = ops.KernelTimer()
    try:
        for v in (-1, 12, 0):
            lib.gdl_debug_force_conv_variant(v)
            for odt in (torch.float32, torch.bfloat16):
                outs[v, odt] = ops.conv_gemm(xn, wq, R=R, S=R, pad=R // 2, bias=bias.to(DEV), act=ops.ACT_RELU,
                                             resid=rn if odt == torch.bfloat16 else rn.float(), out_dtype=odt)
    finally:
        lib.gdl_debug_force_conv_variant(-1)
        ops.TIMER = None
    assert timer.variants == [12, 12, 12, 12, 0, 0], timer.variants        # the planner's own choice for these shapes is the four-stage tile
    close(outs[12, torch.float32].permute(0, 3, 1, 2), ref, dtype, "four-stage tile")
    for odt in (torch.float32, torch.bfloat16):
        assert torch.equal(outs[12, odt], outs[0, odt]), f"four-stage vs two-stage 64^2 tile differ ({odt})"
        assert torch.equal(outs[-1, odt], outs[0, odt])


@pytest.mark.parametrize("B,H,W,C,N,R,expect", [(8, 72, 72, 768, 256, 1, True),     # 1x1, 162 x 1 tiles of 256^2 -> 128^2 tile
                                              (32, 72, 72, 256, 256, 1, True),    # 648 tiles: persistent 256^2 tile
                                              (8, 72, 72, 256, 256, 3, True),     # 3x3
                                              (8, 36, 36, 768, 768, 3, True),     # K = 6912, 81 x 6 tiles of 128^2
                                              (32, 36, 36, 768, 256, 1, True),    # 324 x 2 tiles of 128^2
                                              (2, 36, 36, 768, 256, 1, False),    # too few tiles: 64^2 kernel, register epilogue
                                              (1, 25, 25, 256, 256, 1, False),    # M = 625: no whole tiles -> no statistics
                                              (2, 36, 36, 256, 200, 1, False)])   # N tail
def test_conv_gemm_emits_batchnorm_partial_statistics(B, H, W, C, N, R, expect):
    """conv_gemm(want_stats=True): the train-mode BatchNorm statistics of a ConvModule (models/utils.py:10-52) as a side output of
    the convolution's epilogue -- per-wave partial sums of the bf16-ROUNDED outputs, reduced by bn_stats_finalize: same output
    bits as the plain call, mean / biased variance equal to bn_stats on that output, running statistics updated alike."""
    dtype = torch.bfloat16
    x = q(rnd(B, H, W, C), dtype).to(DEV, dtype)
    w = (q(rnd(N, R * R * C, seed=1), dtype) * 0.05).to(DEV, dtype)
    bias = rnd(N, seed=2).to(DEV)
    y0 = ops.conv_gemm(x, w, R=R, S=R, pad=R // 2, bias=bias)
    y, partials, rows = ops.conv_gemm(x, w, R=R, S=R, pad=R // 2, bias=bias, want_stats=True)
    assert torch.equal(y, y0)
    assert (rows > 0) == expect, rows
    if not rows:
        assert partials is None
        return
    rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    rm2, rv2 = rm.clone(), rv.clone()
    mean, var = ops.bn_stats_finalize(partials, rows, N, B * H * W, rm, rv, 0.1)
    mean_r, var_r = ops.bn_stats(y0, rm2, rv2, 0.1)
    yf = y0.float().reshape(-1, N).cpu().double()
    close(mean.cpu(), yf.mean(0).float(), torch.float32, "mean vs f64 of the stored outputs")
    close(var.cpu(), yf.var(0, unbiased=False).float(), torch.float32, "variance vs f64 of the stored outputs")
    assert (mean - mean_r).abs().max().item() <= 1e-5 * mean_r.abs().max().item() + 1e-6
    assert (var - var_r).abs().max().item() <= 1e-4 * var_r.abs().max().item()
    assert (rm - rm2).abs().max().item() <= 1e-6 and (rv - rv2).abs().max().item() <= 1e-5


@pytest.mark.parametrize("M,N,K", [(10300, 2048, 192), (66000, 512, 64), (20500, 1024, 256), (300, 256, 128)])
def test_conv_persistent_tile(M, N, K):
    """The persistent 256^2 tile (variant 9, conv_gemm_persist.hip: one workgroup per CU walks several tiles, the next tile's first
    stage is in flight under the epilogue): more tiles than CUs (328 / 516 / 324) so that workgroups take a second and third
    tile, odd / single / even K-step counts, an M tail, bf16 and f32 + residual outputs -- vs the 128^2 tile (same K order:
    identical sums), and into a channel slice of a wider buffer."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    dtype = torch.bfloat16
    x, w = q(rnd(M, K), dtype).to(DEV, dtype).view(1, 1, M, K), (q(rnd(N, K, seed=1), dtype) * 0.1).to(DEV, dtype)
    bias, scale = rnd(N, seed=2).to(DEV), rnd(N, seed=3).to(DEV)
    resid = rnd(M, N, seed=4).to(DEV).view(1, 1, M, N)
    outs = {}
    try:
        for v in (9, 1):
            lib.gdl_debug_force_conv_variant(v)
            outs[v, "bf16"] = ops.conv_gemm(x, w, bias=bias, act=ops.ACT_RELU, out_dtype=torch.bfloat16)
            outs[v, "f32"] = ops.conv_gemm(x, w, bias=bias, scale=scale, resid=resid, out_dtype=torch.float32)
            wide = torch.zeros(1, 1, M, N + 64, device=DEV, dtype=torch.bfloat16)
            ops.conv_gemm(x, w, bias=bias, out=wide[..., 32:32 + N])
            outs[v, "slice"] = wide
    finally:
        lib.gdl_debug_force_conv_variant(-1)
    ref = F.relu(x.view(M, K).float().cpu() @ w.float().cpu().t() + bias.cpu())
    close(outs[9, "bf16"].view(M, N), ref, dtype, "persistent tile")
    for key in ("bf16", "f32", "slice"):
        assert torch.equal(outs[9, key], outs[1, key]), f"persistent tile vs 128^2 tile differ ({key})"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,groups,stride", [(256, 32, 1), (512, 32, 1), (1024, 32, 1), (2048, 32, 1), (256, 32, 2), (128, 32, 1)])
def test_grouped_conv_as_batched_supergroups(dtype, C, groups, stride):
    """ResNeXt's grouped 3x3 (torchvision Bottleneck.conv2; the reference's shipped UNet++ encoder is resnext101_32x8d,
    configs/unetplus_config_RGB.yaml:37) as one batched implicit-GEMM launch over super-groups of >= 32 channels
    (gdlhip.cnn.supergroups / ops.conv_gemm_grouped): 8-, 16-, 32- and 64-channel groups, stride 1 and 2 -- forward, data gradient
    and the parameter's gradient against F.conv2d(groups=...) and its autograd, through the training-mode Conv -> BatchNorm -> ReLU
    node that the encoders use, and against the block-diagonal DENSE form of round 3 (GDL_GROUPED_DENSE)."""
    from gdlhip import cnn
    torch.manual_seed(3)
    b, h, w_ = 2, 20, 24
    conv = torch.nn.Conv2d(C, C, 3, padding=1, stride=stride, groups=groups, bias=False)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        conv.weight.copy_(q(conv.weight * 2.0, dtype))
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x = q(rnd(b, C, h, w_, seed=5), dtype)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(bn(conv(xr)))
    gy = rnd(*yr.shape, seed=6)
    yr.backward(gy)
    conv_g = torch.nn.Conv2d(C, C, 3, padding=1, stride=stride, groups=groups, bias=False).to(DEV)
    bn_g = torch.nn.BatchNorm2d(C).to(DEV)
    conv_g.load_state_dict(conv.state_dict()); bn_g.load_state_dict({k: v for k, v in bn.state_dict().items()})
    bn_g.running_mean.zero_(); bn_g.running_var.fill_(1.0); bn_g.num_batches_tracked.zero_()
    cnn.mark_groups(conv_g.weight, groups)
    outs = {}
    for dense in (False, True):
        cnn.GROUPED_BATCHED = not dense
        try:
            conv_g.weight.grad = None
            xg = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_(True)
            sg = cnn.supergroups(conv_g.weight, C, C, dtype)
            assert (sg is None) == dense
            if sg is not None:
                cg = C // groups
                assert sg == (C // max(32, cg), max(1, 32 // cg), max(32, cg), max(32, cg))
            y = cnn.conv_bn(xg, conv_g.weight, bn_g, stride=stride, pad=1, relu=True)
            y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype))
            outs[dense] = (y.detach().float().cpu().permute(0, 3, 1, 2), xg.grad.float().cpu().permute(0, 3, 1, 2),
                           conv_g.weight.grad.float().cpu().clone())
        finally:
            cnn.GROUPED_BATCHED = True
    # f32: both forms against torch autograd.  bf16: the forward against torch; the gradients of a conv -> train-mode BatchNorm ->
    # ReLU chain on two small images move by several per cent with the bf16 rounding of the conv output alone (ReLU mask flips,
    # the mean subtraction of BatchNorm's backward), for either form -- there the batched form is held to the dense one
    rel = lambda a_, r_: ((a_ - r_).norm() / (r_.norm() + 1e-30)).item()      # noqa: E731
    for dense in (False, True):
        y, dx, dw = outs[dense]
        close(y, yr.detach(), dtype, f"grouped conv + BN + ReLU (dense={dense})")
        assert dw.shape == conv.weight.grad.shape
        if dtype == torch.float32:
            close(dx, xr.grad, dtype, f"grouped conv data gradient (dense={dense})", scale=xr.grad.abs().max().item())
            close(dw, conv.weight.grad, dtype, f"grouped conv parameter gradient (dense={dense})", scale=conv.weight.grad.abs().max().item())
        else:
            assert rel(dx, xr.grad.detach()) < 0.1 and rel(dw, conv.weight.grad) < 0.1, (rel(dx, xr.grad.detach()), rel(dw, conv.weight.grad))
    tight = 1e-5 if dtype == torch.float32 else 3e-2
    assert rel(outs[False][0], outs[True][0]) < tight and rel(outs[False][1], outs[True][1]) < tight and rel(outs[False][2], outs[True][2]) < tight


@pytest.mark.parametrize("B,T,N,K", [(26, 1297, 1024, 768), (13, 2600, 512, 1024), (1, 512, 256, 768), (32, 2048, 768, 768)])
def test_conv_parked_tile(B, T, N, K):
    """The persistent one-wave-per-SIMD tile whose finished tile is parked in registers and stored from the MFMA shadows of the next
    tile's K loop (variant 10, conv_gemm_w4p.hip): more tiles than CUs (528 / 266 / 768: workgroups take a second and third tile; and
    one call of two tiles), 12 / 16 K-steps (the "exactly twelve" and the "more" instantiations), an M tail whose rows must be
    dropped by the store descriptor's range check, bias-only outputs, folded BatchNorm + ReLU, output into a channel slice of a
    wider buffer -- all bit-identical to the 128^2 tile -- and BatchNorm partial statistics from phase 1 through the planner."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    dtype = torch.bfloat16
    M = B * T
    x = q(rnd(M, K), dtype).to(DEV, dtype).view(B, 1, T, K)
    w = (q(rnd(N, K, seed=1), dtype) * 0.1).to(DEV, dtype)
    bias, scale, shift = rnd(N, seed=2).to(DEV), rnd(N, seed=3).to(DEV), rnd(N, seed=5).to(DEV)
    outs = {}
    try:
        for v in (10, 1):
            lib.gdl_debug_force_conv_variant(v)
            outs[v, "bf16"] = ops.conv_gemm(x, w, bias=bias)
            outs[v, "nobias"] = ops.conv_gemm(x, w)
            outs[v, "bnrelu"] = ops.conv_gemm(x, w, bias=bias, scale=scale, shift=shift, act=ops.ACT_RELU)
            wide = torch.zeros(B, 1, T, N + 64, device=DEV, dtype=dtype)
            ops.conv_gemm(x, w, bias=bias, out=wide[..., 32:32 + N])
            outs[v, "slice"] = wide
    finally:
        lib.gdl_debug_force_conv_variant(-1)
    pre = x.view(M, K).float().cpu() @ w.float().cpu().t() + bias.cpu()
    close(outs[10, "bf16"].view(M, N), pre, dtype, "parked tile")
    close(outs[10, "bnrelu"].view(M, N), F.relu(pre * scale.cpu() + shift.cpu()), dtype, "parked tile, folded BN + ReLU")
    for key in [k2 for (v, k2) in outs if v == 10]:
        assert torch.equal(outs[10, key], outs[1, key]), f"parked tile vs 128^2 tile differ ({key})"
    if M % 256 == 0 and (M // 256) * (N // 256) >= 512:      # statistics follow the planner's own choice (forced variants emit none)
        lib.gdl_debug_set_conv_w4p.argtypes = [ctypes.c_int]
        lib.gdl_debug_set_conv_w4p(1)                        # (the parked tile is opt-in)
        try:
            y, part, rows = ops.conv_gemm(x, w, bias=bias, want_stats=True)
        finally:
            lib.gdl_debug_set_conv_w4p(0)
        assert rows == M // 128, "the planner should have picked a 256^2 tile with 128-pixel partial rows"
        assert torch.equal(y, outs[1, "bf16"])
        yb = y.view(M, N).double().cpu()
        ref_s = torch.stack([yb.sum(0), (yb * yb).sum(0)])
        got = part.double().sum(0).cpu()
        assert ((got - ref_s).abs() <= 1e-4 * ref_s.abs().max()).all(), "BatchNorm partial statistics from phase 1"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C,N,R", [(2, 37, 23, 64, 64, 3), (1, 50, 50, 128, 32, 1), (3, 16, 16, 192, 64, 3),
                                         # C divides the K chunk: several filter taps are packed into one chunk
                                         (2, 37, 23, 32, 16, 3), (1, 20, 20, 16, 16, 3), (2, 9, 9, 8, 24, 3),
                                         (1, 30, 11, 16, 64, 5)])
def test_conv_narrow_output_tile(dtype, B, H, W, C, N, R):
    """256 x 64 tile for N <= 64 layers (forced here; chosen automatically on large maps): M tails, N tails,
    bias + ReLU + residual epilogue, tap packing for C in {8, 16, 32}, vs F.conv2d and vs the 64^2 tile."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, R, R, seed=1) * 0.1, dtype)
    bias, resid = rnd(N, seed=2), q(rnd(B, N, H, W, seed=3), dtype)
    ref = F.relu(F.conv2d(x, w, bias, padding=R // 2)) + resid
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wq = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV, dtype)
    rn = resid.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    outs = {}
    try:
        for v in (5, 0):
            lib.gdl_debug_force_conv_variant(v)
            outs[v] = ops.conv_gemm(xn, wq, R=R, S=R, pad=R // 2, bias=bias.to(DEV), act=ops.ACT_RELU, resid=rn,
                                    out_dtype=torch.float32)
    finally:
        lib.gdl_debug_force_conv_variant(-1)
    close(outs[5].permute(0, 3, 1, 2), ref, dtype, "256x64 tile")
    assert (outs[5] - outs[0]).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("training", [True, False])
def test_pyramid_fuse_matches_concat_conv(dtype, training):
    """conv1x1(cat(upsampled levels)) + BN + ReLU evaluated per level (gdlhip.nn.pyramid_fuse_bn_act, SegFormer's
    linear_fuse, segformer_mlp.py:97-125) vs torch's interpolate -> cat -> conv2d -> batch_norm -> relu: output, running
    statistics and every gradient (levels, 1x1 weight, BN affine); and vs the build's own unfused path."""
    from torch import nn
    from gdlhip import nn as gnn
    B, E, N = 2, 64, 256
    sizes = [(3, 5), (6, 10), (12, 20), (24, 40)]
    lv = [q(rnd(B, E, h, w, seed=10 + i), dtype) for i, (h, w) in enumerate(sizes)]
    conv = nn.Conv2d(4 * E, N, 1, bias=False)
    bn = nn.BatchNorm2d(N)
    with torch.no_grad():
        conv.weight.copy_(q(rnd(N, 4 * E, 1, 1, seed=1) * 0.1, dtype))
        bn.weight.copy_(rnd(N, seed=2).abs() + 0.5)
        bn.bias.copy_(rnd(N, seed=3) * 0.1)
        bn.running_mean.copy_(rnd(N, seed=4) * 0.1)
        bn.running_var.copy_(rnd(N, seed=5).abs() + 0.5)
    conv.train(training), bn.train(training)
    gout = rnd(B, N, *sizes[-1], seed=7)
    # torch reference (f32, CPU)
    ref_in = [t.clone().requires_grad_(training) for t in lv]
    cat = torch.cat([F.interpolate(t, size=sizes[-1], mode="bilinear", align_corners=False) for t in ref_in[:-1]] + [ref_in[-1]], 1)
    import copy
    rconv, rbn = copy.deepcopy(conv), copy.deepcopy(bn)
    ref = F.relu(rbn(rconv(cat)))
    if training:
        ref.backward(gout)
    outs = {}
    for fused in (True, False):
        gnn.FUSE_PYRAMID = fused
        try:
            c, n = copy.deepcopy(conv).to(DEV).to(memory_format=torch.channels_last), copy.deepcopy(bn).to(DEV)
            xs = [t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_(training) for t in lv]
            with torch.set_grad_enabled(training):
                y = gnn.pyramid_fuse_bn_act(xs, c, n, relu=True)
            if training:
                y.backward(gout.permute(0, 2, 3, 1).contiguous().to(DEV, dtype))
            outs[fused] = (y, xs, c, n)
        finally:
            gnn.FUSE_PYRAMID = True
    y, xs, c, n = outs[True]
    close(y.permute(0, 3, 1, 2), ref, dtype, "pyramid fuse output")
    close(y, outs[False][0], dtype, "fused vs concat path")
    if training:
        close(n.running_mean, rbn.running_mean, dtype, "running_mean")
        close(n.running_var, rbn.running_var, dtype, "running_var")
        def grad_ok(got, ref, other, what):
            """f32: element-wise.  bf16: conv output, dy and the resized dz are all stored in bf16 -- a gradient is held in
            the L2 norm to the f32 reference, no worse than 1.5 x the error of the build's own concat path, and the two
            paths agree with each other."""
            got, ref, other = got.float().cpu(), ref.float().cpu(), other.float().cpu()
            if dtype == torch.float32:
                close(got, ref, dtype, what, scale=ref.abs().max().item())
                return
            rel, rel_other = float((got - ref).norm() / ref.norm()), float((other - ref).norm() / ref.norm())
            rel_paths = float((got - other).norm() / other.norm())
            print(f"{what}: fused vs f32 reference {rel:.4f}, concat path vs reference {rel_other:.4f}, fused vs concat {rel_paths:.4f}")
            assert rel <= max(2e-2, 1.5 * rel_other) and rel_paths <= 6e-2, (what, rel, rel_other, rel_paths)

        _, xs_u, c_u, n_u = outs[False]
        for i, (x, r, xu) in enumerate(zip(xs, ref_in, xs_u)):
            grad_ok(x.grad.permute(0, 3, 1, 2), r.grad, xu.grad.permute(0, 3, 1, 2), f"level {i} gradient")
        grad_ok(c.weight.grad, rconv.weight.grad, c_u.weight.grad, "1x1 weight gradient")
        grad_ok(n.weight.grad, rbn.weight.grad, n_u.weight.grad, "gamma gradient")
        grad_ok(n.bias.grad, rbn.bias.grad, n_u.bias.grad, "beta gradient")



@pytest.mark.parametrize("B,H,W,C,N", [(2, 8, 64, 16, 16), (1, 12, 128, 32, 32), (2, 4, 64, 8, 16), (1, 8, 64, 16, 5),
                                       (1, 16, 192, 32, 16), (3, 4, 64, 16, 32), (1, 8, 64, 32, 320), (2, 4, 64, 16, 72)])
def test_conv3x3_narrow_direct_kernel(B, H, W, C, N):
    """The direct 3x3 kernel for narrow layers on large maps (C in {8,16,32}, outputs in 32-channel slices; chosen by the planner
    whenever it applies): image borders (hardware zero fill of the staged window), several tiles per image and per batch, N tails
    (5 classes), bias + ReLU + residual epilogue with f32 and bf16 outputs, vs F.conv2d and vs the implicit-GEMM tile."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_set_conv_narrow.argtypes = [ctypes.c_int]
    lib.gdl_conv_gemm_plan.restype = ctypes.c_int
    dtype = torch.bfloat16
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, 3, 3, seed=1) * 0.1, dtype)
    bias, resid = rnd(N, seed=2), q(rnd(B, N, H, W, seed=3), dtype)
    ref = F.relu(F.conv2d(x, w, bias, padding=1)) + resid
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wq = w.permute(0, 2, 3, 1).reshape(N, -1).contiguous().to(DEV, dtype)
    rn = resid.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    outs = {}
    try:
        for on in (1, 0):
            lib.gdl_debug_set_conv_narrow(on)
            outs[on] = ops.conv_gemm(xn, wq, R=3, S=3, pad=1, bias=bias.to(DEV), act=ops.ACT_RELU, resid=rn,
                                     out_dtype=torch.float32)
    finally:
        lib.gdl_debug_set_conv_narrow(1)
    close(outs[1].permute(0, 3, 1, 2), ref, dtype, "direct narrow 3x3")
    assert (outs[1] - outs[0]).abs().max().item() <= 1e-4 * ref.abs().max().item()    # same bf16 products, other K order
    yb = ops.conv_gemm(xn, wq, R=3, S=3, pad=1)                                         # plain bf16 output
    close(yb.permute(0, 3, 1, 2), F.conv2d(x, w, padding=1), dtype, "direct narrow 3x3, bf16 out")


@pytest.mark.parametrize("dt_in,dt_out", [(torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32),
                                          (torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32)])
def test_copy_cast_strided(dt_in, dt_out):
    """Strided NHWC copy + dtype conversion: channel slices of wider buffers on both sides, 8- and 4-channel vectors."""
    B, H, W = 2, 5, 7
    for C, Cw in ((16, 48), (12, 20)):
        src = rnd(B, H, W, Cw, seed=C).to(dt_in)
        sd = src.to(DEV)
        x = sd[..., 4:4 + C]
        y = ops.copy_cast(x, out_dtype=dt_out)
        assert y.is_contiguous() and y.dtype == dt_out
        assert torch.equal(y.cpu(), src[..., 4:4 + C].to(dt_out))
        wide = torch.zeros(B, H, W, Cw + 8, device=DEV, dtype=dt_out)
        ops.copy_cast(x, out=wide[..., 8:8 + C])
        assert torch.equal(wide[..., 8:8 + C].cpu(), src[..., 4:4 + C].to(dt_out))
        assert float(wide[..., :8].abs().max()) == 0.0 and float(wide[..., 8 + C:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_concat_resize_conv_backward_per_level(dtype):
    """UperNet's fpn_bottleneck (3x3 over cat([l0, up2(l1), up4(l2), up8(l3)]), upernet.py:144-152): the training node that
    computes the upsampled levels' data and weight gradients at their own resolution, vs torch autograd on the CPU
    (interpolate -> cat -> conv2d -> batch_norm -> relu) and vs the build's concat path."""
    import copy
    from torch import nn
    B, C, N, H, W = 2, 64, 128, 16, 24
    sizes = [(H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    lv = [q(rnd(B, C, h, w, seed=20 + i), dtype) for i, (h, w) in enumerate(sizes)]
    conv_r, bn_r = nn.Conv2d(4 * C, N, 3, padding=1, bias=False), nn.BatchNorm2d(N)
    with torch.no_grad():
        conv_r.weight.copy_(q(rnd(N, 4 * C, 3, 3, seed=1) * 0.05, dtype))
        bn_r.weight.copy_(rnd(N, seed=2).abs() + 0.5)
        bn_r.bias.copy_(rnd(N, seed=3) * 0.1)
    ref_in = [t.clone().requires_grad_() for t in lv]
    cat = torch.cat([ref_in[0]] + [F.interpolate(t, size=(H, W), mode="bilinear", align_corners=False) for t in ref_in[1:]], 1)
    ref = F.relu(bn_r(conv_r(cat)))
    gout = q(rnd(*ref.shape, seed=7), dtype)
    ref.backward(gout)
    outs = {}
    for fused in (True, False):
        gnn.FUSE_CONCAT_BWD = fused
        try:
            c, n = copy.deepcopy(conv_r).to(DEV).to(memory_format=torch.channels_last), copy.deepcopy(bn_r).to(DEV)
            n.reset_running_stats()
            for p_ in list(c.parameters()) + list(n.parameters()):
                p_.grad = None
            xs = [t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_() for t in lv]
            y = gnn.concat_resize_conv_bn_act(xs, c, n.train(), relu=True)
            y.backward(gout.permute(0, 2, 3, 1).contiguous().to(DEV, dtype))
            outs[fused] = (y, xs, c, n)
        finally:
            gnn.FUSE_CONCAT_BWD = True
    y, xs, c, n = outs[True]
    _, xs_u, c_u, n_u = outs[False]
    close(y.permute(0, 3, 1, 2), ref, dtype, "output")

    def grad_ok(got, ref_, other, what):
        got, ref_, other = got.float().cpu(), ref_.float().cpu(), other.float().cpu()
        if dtype == torch.float32:
            close(got, ref_, dtype, what, scale=ref_.abs().max().item())
            return
        rel, rel_other = float((got - ref_).norm() / ref_.norm()), float((other - ref_).norm() / ref_.norm())
        print(f"{what}: per-level backward vs f32 reference {rel:.4f}, concat path vs reference {rel_other:.4f}")
        assert rel <= max(2e-2, 1.5 * rel_other), (what, rel, rel_other)

    for i, (x, r, xu) in enumerate(zip(xs, ref_in, xs_u)):
        grad_ok(x.grad.permute(0, 3, 1, 2), r.grad, xu.grad.permute(0, 3, 1, 2), f"level {i} gradient")
    grad_ok(c.weight.grad, conv_r.weight.grad, c_u.weight.grad, "3x3 weight gradient")
    grad_ok(n.weight.grad, bn_r.weight.grad, n_u.weight.grad, "gamma gradient")
    grad_ok(n.bias.grad, bn_r.bias.grad, n_u.bias.grad, "beta gradient")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Hi,Wi,N,f", [(2, 5, 7, 64, 4), (1, 3, 4, 128, 8), (2, 9, 6, 64, 2), (1, 36, 36, 64, 4), (1, 6, 10, 192, 2),
                                         (2, 1, 1, 64, 4), (1, 2, 13, 128, 4), (1, 72, 72, 64, 2)])
def test_resize_conv3x3_bwd_gather_two_pass(dtype, B, Hi, Wi, N, f):
    """The nine gathered maps G_t = resize^T shift_t^T dy: the one-pass matrix-core kernel (bf16, factors 2 and 4), the separable
    two-pass kernels (rows, then columns) and the single-pass VALU kernel, all vs the definition through torch autograd (d/dx of
    <dy, shift_t(interpolate(x))>)."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_set_gather_mfma.argtypes = [ctypes.c_int]
    Ho, Wo = f * Hi, f * Wi
    dy = q(rnd(B, Ho, Wo, N, seed=3), dtype)
    dyd = dy.to(DEV, dtype)
    outs = {}
    try:
        one_pass = lib.gdl_resize_conv3x3_bwd_gather_one_pass(ops.dt(dyd), B, Ho, Wo, N, Hi, Wi)
        assert bool(one_pass) == (dtype == torch.bfloat16 and f in (2, 4))
        if one_pass:
            outs["mfma"] = ops.resize_conv3x3_bwd_gather(dyd, (Hi, Wi)).float().cpu()
        lib.gdl_debug_set_gather_mfma(0)
        for two in (True, False):
            ops.GATHER_TWO_PASS = two
            outs[two] = ops.resize_conv3x3_bwd_gather(dyd, (Hi, Wi)).float().cpu()
    finally:
        ops.GATHER_TWO_PASS = True
        lib.gdl_debug_set_gather_mfma(1)
    # definition: G_t[q, n] = d/dx[q] sum_p dy[p, n] * (shift_t(U x))[p], one channel at a time is independent -> use x = ones-probe
    x = torch.zeros(B, N, Hi, Wi, requires_grad=True)
    up = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
    padded = F.pad(up, (1, 1, 1, 1))
    dyn = dy.permute(0, 3, 1, 2)
    ref = torch.empty(B, Hi, Wi, 9 * N)
    for r in range(3):
        for s3 in range(3):
            (gx,) = torch.autograd.grad((padded[:, :, r:r + Ho, s3:s3 + Wo] * dyn).sum(), x, retain_graph=True)
            t = 3 * r + s3
            ref[..., (8 - t) * N:(9 - t) * N] = gx.permute(0, 2, 3, 1)
    close(outs[False], ref, dtype, "single-pass gather")
    close(outs[True], ref, dtype, "two-pass gather")
    if "mfma" in outs:
        close(outs["mfma"], ref, dtype, "one-pass matrix-core gather")
        print(f"max |error|: one-pass {(outs['mfma'] - ref).abs().max().item():.3e}, two-pass {(outs[True] - ref).abs().max().item():.3e}")
    if dtype == torch.float32:
        assert (outs[True] - outs[False]).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("B,Hi,Wi,N,f", [(2, 5, 7, 64, 4), (2, 9, 6, 128, 2), (1, 36, 36, 192, 4), (2, 1, 1, 64, 4), (1, 2, 13, 128, 4),
                                         (1, 37, 18, 64, 2)])
def test_resize_conv3x3_bwd_gather_with_batchnorm_backward_fused(B, Hi, Wi, N, f, relu):
    """Round 5: gdl_resize_conv3x3_bwd_gather_bn = gather(bn_bwd_dx(y, dz)) in one kernel (the BatchNorm + ReLU backward is applied
    while dz is staged; its result never exists in memory).  Against (a) the two launches it replaces, which round the same f32
    values to bf16 at the same place (only the order of the f32 operations differs), and (b) the definition in f32 torch:
    autograd of relu(batch_norm(y)) contracted with dz, gathered through the transposed resize + tap shifts."""
    dtype = torch.bfloat16
    Ho, Wo = f * Hi, f * Wi
    y = q(rnd(B, Ho, Wo, N, seed=5) * 1.5 + rnd(N, seed=6) * 0.7, dtype)             # per-channel means far from zero
    dz = q(rnd(B, Ho, Wo, N, seed=7), dtype)
    gamma, beta = rnd(N, seed=8) * 0.8 + 0.2, rnd(N, seed=9) * 0.5                    # some negative gammas
    eps = 1e-5
    yd, dzd, gd, bd = y.to(DEV, dtype), dz.to(DEV, dtype), gamma.to(DEV), beta.to(DEV)
    mean, var = ops.bn_stats(yd)
    dgs, dbs = ops.bn_bwd_reduce(yd, dzd, mean, var, gd, bd, eps, relu)
    P = B * Ho * Wo
    assert ops.resize_conv3x3_bwd_gather_bn_ok(dzd, (Hi, Wi))
    fused = ops.resize_conv3x3_bwd_gather_bn(dzd, yd, (Hi, Wi), mean, var, gd, bd, eps, relu, dgs, dbs, P)
    dy2 = ops.bn_bwd_dx(yd, dzd, mean, var, gd, bd, eps, relu, dgs, dbs, P)
    two = ops.resize_conv3x3_bwd_gather(dy2, (Hi, Wi))
    scale = two.float().abs().max().item()
    diff = (fused.float() - two.float()).abs()
    assert diff.max().item() <= 2e-2 * scale, (diff.max().item(), scale)
    assert (diff > 4e-3 * scale).float().mean().item() < 2e-2           # bf16 rounding of a few dy elements, not a different formula
    # definition (f32, CPU)
    yr = y.clone().requires_grad_(True)
    out = F.batch_norm(yr.permute(0, 3, 1, 2), None, None, gamma, beta, True, 0.1, eps)
    out = F.relu(out) if relu else out
    (dy_ref,) = torch.autograd.grad(out, yr, dz.permute(0, 3, 1, 2))
    x = torch.zeros(B, N, Hi, Wi, requires_grad=True)
    padded = F.pad(F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False), (1, 1, 1, 1))
    dyn = dy_ref.permute(0, 3, 1, 2)
    ref = torch.empty(B, Hi, Wi, 9 * N)
    for r in range(3):
        for s3 in range(3):
            (gx,) = torch.autograd.grad((padded[:, :, r:r + Ho, s3:s3 + Wo] * dyn).sum(), x, retain_graph=True)
            t = 3 * r + s3
            ref[..., (8 - t) * N:(9 - t) * N] = gx.permute(0, 2, 3, 1)
    # ReLU decisions at |bn(y)| ~ 1e-7 may differ between two f32 evaluations: bound the share, hold the rest to bf16
    err = (fused.float().cpu() - ref).abs()
    rs = ref.abs().max().item()
    assert (err > 2e-2 * rs).float().mean().item() < 1e-3, (err.max().item(), rs)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mfma", [1, 5, 2, 4, 0])
@pytest.mark.parametrize("B,H,W,N,factors,vec", [(2, 16, 24, 64, (2,), 0), (1, 36, 36, 128, (4,), 0), (2, 16, 16, 64, (8,), 0),
                                                 (2, 16, 24, 64, (2, 4, 8), 0), (1, 8, 8, 72, (4, 2), 0), (2, 12, 20, 64, (4,), 4),
                                                 (1, 2, 2, 64, (2,), 0), (1, 18, 14, 64, (2,), 0), (1, 8, 72, 128, (4,), 0),
                                                 (2, 32, 48, 192, (8, 4), 0), (1, 144, 144, 64, (2, 4, 8), 0),
                                                 (4, 240, 256, 64, (4,), 0), (3, 160, 192, 128, (8,), 0),    # 3840 / 1440 patches: persistent workgroups take several
                                                 (4, 32, 272, 384, (2, 4, 8), 0), (1, 8, 16, 64, (2, 4, 8), 0)])   # three rolling rings: several columns per workgroup; one cell of the coarsest level
def test_resize_conv3x3_fwd_sum(dtype, B, H, W, N, factors, vec, mfma):
    """sum_k sum_t shift_t(bilinear(z_k,t)) (gdl_resize_conv3x3_fwd_sum), the pixel side of the low-resolution forward of
    conv3x3(pad 1)(bilinear resize(x)) (multilevel_neck.py:157-158, upernet.py:144-152): vs torch's interpolate / pad / slice
    on the CPU, every pixel including the border lines the zero padding touches; with the bias / folded-BN + ReLU epilogue.
    mfma: 1 = the matrix-core kernels with the production choice of version, 5 = version 2 (bf16, N % 64 == 0: all channels of
    a 4 x 16 pixel block per workgroup, pipelined windows, channel-independent state in registers), 2 / 4 = version 1 (32- / 16-column blocks x 64 channels), 0 = the pixel-by-pixel kernel
    (also what f32 and other channel counts run)."""
    import ctypes
    from gdlhip import _lib
    if mfma != 1 and (dtype != torch.bfloat16 or N % 64 or vec):
        pytest.skip("the matrix-core variants only exist for bf16 with N % 64 == 0")
    lib = _lib.load()
    lib.gdl_debug_set_tapsum_vec.argtypes = [ctypes.c_int]
    lib.gdl_debug_set_tapsum_mfma.argtypes = [ctypes.c_int]
    zs = [q(rnd(B, H // f, W // f, 9 * N, seed=10 + f), dtype) for f in factors]
    ref = torch.zeros(B, N, H, W)
    for z in zs:
        zt = z.view(B, z.shape[1], z.shape[2], 9, N).permute(0, 3, 4, 1, 2)            # [B, 9, N, h, w]
        for t in range(9):
            up = F.pad(F.interpolate(zt[:, t], size=(H, W), mode="bilinear", align_corners=False), (1, 1, 1, 1))
            ref += up[:, :, t // 3:t // 3 + H, t % 3:t % 3 + W]
    ref = ref.permute(0, 2, 3, 1)
    add = rnd(N, seed=5)
    lib.gdl_debug_set_tapsum_vec(vec)
    lib.gdl_debug_set_tapsum_mfma(0 if vec else mfma)
    try:
        y = ops.resize_conv3x3_fwd_sum([z.to(DEV, dtype) for z in zs], (H, W))
        y2 = ops.resize_conv3x3_fwd_sum([z.to(DEV, dtype) for z in zs], (H, W), addvec=add.to(DEV), relu=True)
    finally:
        lib.gdl_debug_set_tapsum_vec(0)
        lib.gdl_debug_set_tapsum_mfma(1)
    close(y, ref, dtype, "tap sum")
    for name, sl in (("top", (slice(None), 0)), ("bottom", (slice(None), -1)), ("left", (slice(None), slice(None), 0)),
                     ("right", (slice(None), slice(None), -1))):
        close(y[sl], ref[sl], dtype, f"tap sum {name} line", scale=ref.abs().max().item())
    close(y2, F.relu(ref + add), dtype, "tap sum + shift + ReLU")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,N,lows", [(2, 73, 73, 64, [(9, 9)]), (1, 40, 56, 64, [(20, 28), (10, 14), (5, 6)]),
                                          (2, 37, 29, 72, [(9, 7), (4, 3)]), (1, 292, 292, 64, [(146, 146), (73, 73), (36, 36)])])
def test_resize_conv3x3_fwd_sum_any_ratio(dtype, B, H, W, N, lows):
    """Round 5: sources whose size does NOT divide the output's (DOFA-large's FPN pyramid is 292 / 146 / 73 / 36:
    models/utils.py:106-110 gives int(73 * 0.5) = 36, i.e. x 8.11) -- integer-factor sources still take the one-pass cell kernels,
    every other source one plain gather pass that accumulates into the same output; addvec / ReLU after the last pass.
    vs torch's interpolate / pad / slice per tap."""
    zs = [q(rnd(B, h, w, 9 * N, seed=20 + h), dtype) for h, w in lows]
    assert not all(ops.resize_conv3x3_fwd_ok((h, w), (H, W), B) for h, w in lows)
    ref = torch.zeros(B, N, H, W)
    for z in zs:
        zt = z.view(B, z.shape[1], z.shape[2], 9, N).permute(0, 3, 4, 1, 2)
        for t in range(9):
            up = F.pad(F.interpolate(zt[:, t], size=(H, W), mode="bilinear", align_corners=False), (1, 1, 1, 1))
            ref += up[:, :, t // 3:t // 3 + H, t % 3:t % 3 + W]
    ref = ref.permute(0, 2, 3, 1)
    add = rnd(N, seed=5)
    zd = [z.to(DEV, dtype) for z in zs]
    y = ops.resize_conv3x3_fwd_sum(zd, (H, W))
    y2 = ops.resize_conv3x3_fwd_sum(zd, (H, W), addvec=add.to(DEV), relu=True)
    close(y, ref, dtype, "tap sum, any ratio")
    for name, sl in (("top", (slice(None), 0)), ("bottom", (slice(None), -1)), ("left", (slice(None), slice(None), 0)),
                     ("right", (slice(None), slice(None), -1))):
        close(y[sl], ref[sl], dtype, f"tap sum {name} line", scale=ref.abs().max().item())
    close(y2, F.relu(ref + add), dtype, "tap sum + shift + ReLU, any ratio")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Hi,Wi,N,Ho,Wo", [(2, 9, 9, 64, 73, 73), (1, 5, 6, 64, 40, 56), (1, 36, 36, 64, 292, 292), (2, 4, 3, 72, 37, 29),
                                             (1, 7, 7, 64, 10, 10)])
def test_resize_conv3x3_bwd_gather_any_ratio(dtype, B, Hi, Wi, N, Ho, Wo):
    """The nine gathered maps G_t = resize^T shift_t^T dy for non-integer resize ratios up to 10 (36 -> 292 needs a window of 22
    columns: the NX = 24 instantiations), single-pass and two-pass kernels, vs the definition through torch autograd."""
    dy = q(rnd(B, Ho, Wo, N, seed=3), dtype)
    dyd = dy.to(DEV, dtype)
    outs = {}
    try:
        for two in (True, False):
            ops.GATHER_TWO_PASS = two
            outs[two] = ops.resize_conv3x3_bwd_gather(dyd, (Hi, Wi)).float().cpu()
    finally:
        ops.GATHER_TWO_PASS = True
    x = torch.zeros(B, N, Hi, Wi, requires_grad=True)
    padded = F.pad(F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False), (1, 1, 1, 1))
    dyn = dy.permute(0, 3, 1, 2)
    ref = torch.empty(B, Hi, Wi, 9 * N)
    for r in range(3):
        for s3 in range(3):
            (gx,) = torch.autograd.grad((padded[:, :, r:r + Ho, s3:s3 + Wo] * dyn).sum(), x, retain_graph=True)
            t = 3 * r + s3
            ref[..., (8 - t) * N:(9 - t) * N] = gx.permute(0, 2, 3, 1)
    close(outs[False], ref, dtype, "single-pass gather, any ratio")
    close(outs[True], ref, dtype, "two-pass gather, any ratio")


@pytest.mark.parametrize("dtype", DTYPES)
def test_concat_resize_conv_with_a_non_integer_level(dtype):
    """UperNet `fpn_bottleneck` (upernet.py:144-152) over a DOFA-large-shaped pyramid in miniature (37 / 18 / 9 / 4: the top
    level's x 9.25 is no integer factor, like 292 / 36): eval and train forward + every gradient through the per-level
    low-resolution forms -- no concat buffer, no `runs UNFUSED` warning -- vs torch autograd of interpolate -> cat -> conv2d ->
    batch_norm -> relu."""
    import copy
    warned = set(gnn._WARNED_FALLBACK)
    B, C, N, sizes = 2, 64, 64, [(37, 37), (18, 18), (9, 9), (4, 4)]
    lv = [q(rnd(B, C, h, w, seed=30 + h), dtype) for h, w in sizes]
    conv_r, bn_r = torch.nn.Conv2d(4 * C, N, 3, padding=1, bias=False), torch.nn.BatchNorm2d(N)
    with torch.no_grad():
        conv_r.weight.copy_(q(rnd(N, 4 * C, 3, 3, seed=1) * 0.05, dtype))
        bn_r.weight.copy_(rnd(N, seed=2).abs() + 0.5)
        bn_r.bias.copy_(rnd(N, seed=3) * 0.1)
    conv, bn = copy.deepcopy(conv_r).to(DEV), copy.deepcopy(bn_r).to(DEV)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)

    def reference(train):
        xs = [t.clone().requires_grad_(True) for t in lv]
        cat = torch.cat([xs[0]] + [F.interpolate(t, size=sizes[0], mode="bilinear", align_corners=False) for t in xs[1:]], 1)
        bn_r.train(train)
        return xs, F.relu(bn_r(conv_r(cat)))

    def ours(train):
        xs = [t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_(True) for t in lv]
        bn.train(train)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            out = gnn.concat_resize_conv_bn_act(xs, conv, bn)
        assert set(gnn._WARNED_FALLBACK) == warned, "the node fell back to its UNFUSED path"
        return xs, out
    with torch.no_grad():
        _, ref_e = reference(False)
        _, got_e = ours(False)
    close(got_e.permute(0, 3, 1, 2), ref_e, dtype, "eval forward")
    xs_r, ref_t = reference(True)
    xs_o, got_t = ours(True)
    close(got_t.permute(0, 3, 1, 2), ref_t, dtype, "train forward")
    go = q(rnd(*ref_t.shape, seed=9), dtype)
    ref_t.backward(go)
    got_t.backward(go.permute(0, 2, 3, 1).contiguous().to(DEV, got_t.dtype))
    gt = 1e-3 if dtype == torch.float32 else 4e-2
    for j, (a, b) in enumerate(zip(xs_o, xs_r)):
        err = (a.grad.float().cpu().permute(0, 3, 1, 2) - b.grad).norm().item()
        assert err <= gt * b.grad.norm().item(), (j, err, b.grad.norm().item())
    err = (conv.weight.grad.float().cpu() - conv_r.weight.grad).norm().item()
    assert err <= gt * conv_r.weight.grad.norm().item(), ("weight", err)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("up", [2, 4])
def test_resized_conv_eval_and_train_forward_at_low_resolution(dtype, up):
    """ConvModule(3x3, bias) on a bilinearly upsampled input (multilevel_neck.py:157-158), forward through the nine
    low-resolution tap products: eval mode (BatchNorm folded into the tap weights / the gather-sum epilogue) and train mode
    (batch statistics incl. the conv bias in running_mean) vs torch's interpolate -> conv2d -> batch_norm -> relu, and vs the
    round-2 forward (gnn.FUSE_TAPSUM = False)."""
    import copy
    B, H, W, C, N = 2, 9, 6, 64, 128
    x, w = q(rnd(B, C, H, W), dtype), q(rnd(N, C, 3, 3, seed=1) * 0.1, dtype)
    conv_r, bn_r = torch.nn.Conv2d(C, N, 3, padding=1, bias=True), torch.nn.BatchNorm2d(N)
    with torch.no_grad():
        conv_r.weight.copy_(w)
        conv_r.bias.copy_(rnd(N, seed=6) * 0.3)
        bn_r.weight.copy_(rnd(N, seed=2).abs() + 0.5)
        bn_r.bias.copy_(rnd(N, seed=3) * 0.1)
        bn_r.running_mean.copy_(rnd(N, seed=4) * 0.2)
        bn_r.running_var.copy_(rnd(N, seed=5).abs() + 0.5)
    conv, norm = copy.deepcopy(conv_r).to(DEV).to(memory_format=torch.channels_last), copy.deepcopy(bn_r).to(DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    up_r = F.interpolate(x, scale_factor=up, mode="bilinear", align_corners=False)
    with torch.no_grad():
        ref_eval = F.relu(bn_r.eval()(conv_r(up_r)))
        got_eval = gnn.conv_bn_act(xd, conv, norm.eval(), relu=True, up=up)
        gnn.FUSE_TAPSUM = False
        try:
            old_eval = gnn.conv_bn_act(xd, conv, norm.eval(), relu=True, up=up)
        finally:
            gnn.FUSE_TAPSUM = True
    scale = None if dtype == torch.float32 else 2 * ref_eval.abs().max().item()
    close(got_eval.permute(0, 3, 1, 2), ref_eval, dtype, "eval output", scale=scale)
    close(got_eval, old_eval, dtype, "eval output vs round-2 forward", scale=scale)
    ref_train = F.relu(bn_r.train()(conv_r(up_r)))
    with torch.no_grad():
        got_train = gnn.conv_bn_act(xd, conv, norm.train(), relu=True, up=up)
    close(got_train.permute(0, 3, 1, 2), ref_train, dtype, "train output", scale=scale)
    close(norm.running_mean, bn_r.running_mean, dtype, "running_mean")
    close(norm.running_var, bn_r.running_var, dtype, "running_var")


@pytest.mark.parametrize("dtype", DTYPES)
def test_concat_resize_conv_eval_per_level(dtype):
    """UperNet's fpn_bottleneck in eval mode (upernet.py:144-152): native level through the 3x3 kernel on its weight slice, the
    x2 / x4 / x8 levels through their low-resolution tap products, folded BatchNorm; vs torch (interpolate -> cat -> conv2d ->
    batch_norm(eval) -> relu) and vs the concat-buffer path (gnn.FUSE_TAPSUM = False)."""
    import copy
    from torch import nn
    B, C, N, H, W = 2, 64, 256, 16, 24
    sizes = [(H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    lv = [q(rnd(B, C, h, w, seed=20 + i), dtype) for i, (h, w) in enumerate(sizes)]
    conv_r, bn_r = nn.Conv2d(4 * C, N, 3, padding=1, bias=False), nn.BatchNorm2d(N)
    with torch.no_grad():
        conv_r.weight.copy_(q(rnd(N, 4 * C, 3, 3, seed=1) * 0.05, dtype))
        bn_r.weight.copy_(rnd(N, seed=2).abs() + 0.5)
        bn_r.bias.copy_(rnd(N, seed=3) * 0.1)
        bn_r.running_mean.copy_(rnd(N, seed=4) * 0.2)
        bn_r.running_var.copy_(rnd(N, seed=5).abs() + 0.5)
        cat = torch.cat([lv[0]] + [F.interpolate(t, size=(H, W), mode="bilinear", align_corners=False) for t in lv[1:]], 1)
        ref = F.relu(bn_r.eval()(conv_r(cat)))
    c, n = copy.deepcopy(conv_r).to(DEV).to(memory_format=torch.channels_last), copy.deepcopy(bn_r).to(DEV).eval()
    xs = [t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype) for t in lv]
    with torch.no_grad():
        y = gnn.concat_resize_conv_bn_act(xs, c, n, relu=True)
        gnn.FUSE_TAPSUM = False
        try:
            y_old = gnn.concat_resize_conv_bn_act(xs, c, n, relu=True)
        finally:
            gnn.FUSE_TAPSUM = True
    scale = None if dtype == torch.float32 else 2 * ref.abs().max().item()
    close(y.permute(0, 3, 1, 2), ref, dtype, "eval output", scale=scale)
    close(y, y_old, dtype, "per-level vs concat path", scale=scale)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_epilogue_v2_bit_identical_to_round2_epilogue(dtype):
    """The round-3 epilogue of the 256^2 tiles (element-wise terms behind the LDS transpose, per-lane channel constants,
    residual rows requested one pass ahead) performs the same operations in the same order per element as the round-2
    one: every combination the models use must come out bit-identical (gdl_debug_set_conv_epilogue switches).  Linear
    layers with M tail and a sample boundary inside a pass, a 3x3 convolution through the shared-staging kernel."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_force_conv_variant.argtypes = [ctypes.c_int]
    lib.gdl_debug_set_conv_epilogue.argtypes = [ctypes.c_int]
    B, T, K, N = 3, 1297, 128, 512
    xd = q(rnd(B, T, K), dtype).to(DEV, dtype).view(B, 1, T, K)
    wd = (q(rnd(N, K, seed=1), dtype) * 0.1).to(DEV, dtype)
    bias, scale, shift = rnd(N, seed=2).to(DEV), rnd(N, seed=3).to(DEV), rnd(N, seed=4).to(DEV)
    bs = torch.tensor([0.0, 1.25, 0.5], device=DEV)
    resid = rnd(B, T, N, seed=5)
    xc = q(rnd(2, 19, 23, 64, seed=7), dtype).to(DEV, dtype)
    wc = (q(rnd(256, 9 * 64, seed=8), dtype) * 0.05).to(DEV, dtype)
    rc = rnd(2, 19, 23, 256, seed=9)
    cases = []
    for odt in DTYPES:
        cases.append((f"plain bias {odt}", lambda odt=odt: ops.conv_gemm(xd, wd, bias=bias, out_dtype=odt)))
        cases.append((f"gelu {odt}", lambda odt=odt: ops.conv_gemm(xd, wd, bias=bias, act=ops.ACT_GELU, out_dtype=odt)))
        cases.append((f"folded BN + relu {odt}", lambda odt=odt: ops.conv_gemm(xd, wd, scale=scale, shift=shift, act=ops.ACT_RELU, out_dtype=odt)))
        for rdt in DTYPES:
            r1 = q(resid, rdt).to(DEV, rdt).view(B, 1, T, N)
            cases.append((f"LayerScale + DropPath + residual {odt} {rdt}",
                          lambda odt=odt, r1=r1: ops.conv_gemm(xd, wd, bias=bias, scale=scale, batch_scale=bs, resid=r1, out_dtype=odt)))
            cases.append((f"residual then relu {odt} {rdt}",
                          lambda odt=odt, r1=r1: ops.conv_gemm(xd, wd, bias=bias, resid=r1, act=ops.ACT_RESID_RELU, out_dtype=odt)))
            r2 = q(rc, rdt).to(DEV, rdt)
            cases.append((f"3x3 + residual {odt} {rdt}",
                          lambda odt=odt, r2=r2: ops.conv_gemm(xc, wc, R=3, S=3, pad=1, resid=r2, out_dtype=odt)))
    lib.gdl_debug_force_conv_variant(3)
    try:
        for name, fn in cases:
            if name.startswith("3x3"):
                lib.gdl_debug_force_conv_variant(4)
            outs = []
            for v2 in (1, 0):
                lib.gdl_debug_set_conv_epilogue(v2)
                outs.append(fn())
            lib.gdl_debug_force_conv_variant(3)
            assert torch.isfinite(outs[0].float()).all(), name
            assert torch.equal(outs[0], outs[1]), f"{name}: max dev {(outs[0].float() - outs[1].float()).abs().max().item():.3e}"
    finally:
        lib.gdl_debug_set_conv_epilogue(1)
        lib.gdl_debug_force_conv_variant(-1)


@pytest.mark.parametrize("B,H,W,N,factors", [(2, 16, 24, 64, (2,)), (1, 36, 36, 128, (4,)), (2, 18, 14, 192, (2,)), (2, 32, 48, 64, (8, 4)),
                                             (4, 240, 256, 64, (4,)), (2, 144, 160, 64, (2, 4, 8))])   # more patches than resident workgroups
def test_resize_conv3x3_fwd_sum_with_batchnorm_statistics(B, H, W, N, factors):
    """gdl_resize_conv3x3_fwd_sum_bn: the gather-sum with train-mode BatchNorm statistics as a side output -- same output
    bits as the plain call, mean / biased variance / running statistics equal to nn.BatchNorm2d's on that (bf16) output."""
    dtype = torch.bfloat16
    zs = [q(rnd(B, H // f, W // f, 9 * N, seed=10 + f), dtype).to(DEV, dtype) for f in factors]
    add = rnd(N, seed=5).to(DEV)
    y0 = ops.resize_conv3x3_fwd_sum(zs, (H, W), addvec=add)
    rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    y, mean, var = ops.resize_conv3x3_fwd_sum_bn(zs, (H, W), addvec=add, running_mean=rm, running_var=rv, momentum=0.1)
    if factors == (2, 4, 8):
        # round 6: the plain call walks three rolling rings (version 3), the statistics variant of three sources is version 2:
        # the same products in another summation order -- one bf16 step apart at most, and rarely
        d = (y.float() - y0.float()).abs()
        assert d.max().item() <= 2.0 ** -7 * y0.float().abs().max().item() and (d > 0).float().mean().item() < 0.02
    else:
        assert torch.equal(y, y0)
    bn = torch.nn.BatchNorm2d(N).train()
    bn(y.float().cpu().permute(0, 3, 1, 2))
    yf = y.float().cpu().reshape(-1, N)
    close(mean, yf.mean(0), torch.float32, "mean", scale=yf.abs().max().item())
    close(var, yf.var(0, unbiased=False), torch.float32, "var")
    close(rm, bn.running_mean, torch.float32, "running_mean", scale=yf.abs().max().item())
    close(rv, bn.running_var, torch.float32, "running_var")


@pytest.mark.parametrize("B,H,W,N,f", [(1, 4, 16, 64, 4), (1, 4, 8, 64, 2), (2, 8, 24, 128, 2), (2, 72, 72, 192, 2), (1, 144, 144, 128, 4),
                                       (5, 16, 512, 512, 4), (3, 8, 488, 1024, 2), (2, 36, 40, 64, 4)])
def test_resize_conv3x3_fwd_sum_rolling_window(B, H, W, N, f):
    """Round 6, version 3 of the forward gather-sum (one source of factor 2 / 4, H % 4 == 0): a workgroup walks a column
    (image, 64 channels, 16 output columns) top to bottom over a ring of low-resolution rows, rows outside the image clamped in
    the DMA address, the zero padding as masked weight fragments in the first and last step, BatchNorm statistics on the matrix
    cores.  Against torch on the CPU (every pixel, border lines separately), against version 2 (same formula, another summation
    order), and the statistics against the output they describe.  Shapes: one step (top and bottom at once), strips that end
    outside the image (W % 16 != 0), more columns than resident workgroup groups (a workgroup walks several)."""
    import ctypes
    from gdlhip import _lib
    lib = _lib.load()
    lib.gdl_debug_set_tapsum_roll.argtypes = [ctypes.c_int]
    dtype = torch.bfloat16
    z = q(rnd(B, H // f, W // f, 9 * N, seed=31 + f), dtype)
    zt = z.view(B, H // f, W // f, 9, N).permute(0, 3, 4, 1, 2)
    ref = torch.zeros(B, N, H, W)
    for t in range(9):
        up = F.pad(F.interpolate(zt[:, t].float(), size=(H, W), mode="bilinear", align_corners=False), (1, 1, 1, 1))
        ref += up[:, :, t // 3:t // 3 + H, t % 3:t % 3 + W]
    ref = ref.permute(0, 2, 3, 1)
    add = rnd(N, seed=5)
    zd = z.to(DEV, dtype)
    y = ops.resize_conv3x3_fwd_sum([zd], (H, W))
    y2 = ops.resize_conv3x3_fwd_sum([zd], (H, W), addvec=add.to(DEV), relu=True)
    rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    y3, mean, var = ops.resize_conv3x3_fwd_sum_bn([zd], (H, W), addvec=add.to(DEV), running_mean=rm, running_var=rv, momentum=0.1)
    lib.gdl_debug_set_tapsum_roll(0)
    try:
        v = ops.resize_conv3x3_fwd_sum([zd], (H, W))
        v3, vmean, vvar = ops.resize_conv3x3_fwd_sum_bn([zd], (H, W), addvec=add.to(DEV))
    finally:
        lib.gdl_debug_set_tapsum_roll(1)
    close(y, ref, dtype, "tap sum")
    for name, sl in (("top", (slice(None), 0)), ("bottom", (slice(None), -1)), ("left", (slice(None), slice(None), 0)),
                     ("right", (slice(None), slice(None), -1))):
        close(y[sl], ref[sl], dtype, f"tap sum {name} line", scale=ref.abs().max().item())
    close(y2, F.relu(ref + add), dtype, "tap sum + shift + ReLU")
    # version 2 sums the same products in another order: one bf16 step apart at most, and only rarely
    dv = (y.float() - v.float()).abs()
    assert dv.max().item() <= 2.0 ** -7 * ref.abs().max().item(), dv.max().item()
    assert (dv > 0).float().mean().item() < 0.02, (dv > 0).float().mean().item()
    d3 = (y3.float() - v3.float()).abs()
    assert (d3 > 0).float().mean().item() < 0.02
    yf = y3.float().cpu().reshape(-1, N)
    close(mean, yf.mean(0), torch.float32, "mean", scale=yf.abs().max().item())
    close(var, yf.var(0, unbiased=False), torch.float32, "var")
    close(mean, vmean, torch.float32, "mean vs version 2", scale=yf.abs().max().item())
    close(var, vvar, torch.float32, "var vs version 2")
    bn = torch.nn.BatchNorm2d(N).train()
    bn(y3.float().cpu().permute(0, 3, 1, 2))
    close(rm, bn.running_mean, torch.float32, "running_mean", scale=yf.abs().max().item())
    close(rv, bn.running_var, torch.float32, "running_var")


def _rel_l2(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-12))


def _resized_conv_nodes_at_production_shape(dtype):
    """The two algebraically restructured nodes of DOFA-base + UperNet at their PRODUCTION shapes (B = 2): the
    neck's x4 level (ConvModule 768 -> 768 on a 36^2 map resized to 144^2, multilevel_neck.py:157-158) and UperNet's
    fpn_bottleneck (1024 -> 256 over [144^2, 72^2, 36^2, 18^2], upernet.py:144-152).  Forward through the nine
    low-resolution tap products, backward through the nine gathered maps; reference = torch f32 autograd of
    interpolate -> (cat ->) conv2d -> batch_norm -> relu on the CPU, fed the SAME (dtype-rounded) operands.
    Two references: (a) the plain f32 graph; (b) the same f32 graph evaluated AT the build's forward value (its conv output
    substituted straight-through, so both sides use the same ReLU mask and batch statistics): what remains is the arithmetic
    of the backward kernels.  Returns {"<node> <tensor> [<reference>]": relative L2 error}."""
    import copy
    from torch import nn
    B = 2
    res = {}

    def run_case(tag, make_input_refs, forward_ref, conv_r, bn_r, run_build):
        """make_input_refs() -> leaf tensors; forward_ref(leaves) -> pre-BatchNorm conv output (f32 graph)."""
        conv, norm = copy.deepcopy(conv_r).to(DEV).to(memory_format=torch.channels_last), copy.deepcopy(bn_r).to(DEV)
        for p_ in list(conv.parameters()) + list(norm.parameters()):
            p_.grad = None
        leaves = make_input_refs()
        yc = forward_ref(leaves)
        gy = q(rnd(*yc.shape, seed=45), dtype)
        y_ours_pre, got = run_build(conv, norm, gy)          # the build's own conv output (NCHW f32 on the CPU) and its results
        for mode in ("f32", "at the build's forward"):
            for p_ in list(conv_r.parameters()) + list(bn_r.parameters()):
                p_.grad = None
            leaves = make_input_refs()
            yc = forward_ref(leaves)
            if mode != "f32":
                yc = yc + (y_ours_pre - yc).detach()
            yr = F.relu(bn_r(yc))
            yr.backward(gy)
            ref = {"out": yr.detach(), "dw": conv_r.weight.grad, "dgamma": bn_r.weight.grad, "dbeta": bn_r.bias.grad,
                   **{f"d input {i}": t.grad for i, t in enumerate(leaves)}}
            for k, v in got.items():
                res[f"{tag} {k} [{mode}]"] = _rel_l2(v, ref[k])

    # ---- neck x4 level
    x, w = q(rnd(B, 768, 36, 36, seed=40), dtype), q(rnd(768, 768, 3, 3, seed=41) * 0.02, dtype)
    conv_r, bn_r = nn.Conv2d(768, 768, 3, padding=1, bias=True), nn.BatchNorm2d(768)
    with torch.no_grad():
        conv_r.weight.copy_(w)
        conv_r.bias.copy_(rnd(768, seed=42) * 0.1)
        bn_r.weight.copy_(rnd(768, seed=43).abs() + 0.5)
        bn_r.bias.copy_(rnd(768, seed=44) * 0.1)

    def build_neck(conv, norm, gy):
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_()
        with torch.no_grad():      # the node's conv output before BatchNorm (same kernels, statistics untouched)
            pre, _ = gnn._cba_conv(xd.detach(), conv.weight, conv.bias.detach(), 1, 4, False, None, None, 0.1)
        y = gnn.conv_bn_act(xd, conv, norm.train(), relu=True, up=4)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype))
        return pre.float().cpu().permute(0, 3, 1, 2), {"out": y.detach().permute(0, 3, 1, 2), "d input 0": xd.grad.permute(0, 3, 1, 2),
                                                      "dw": conv.weight.grad, "dgamma": norm.weight.grad, "dbeta": norm.bias.grad}

    run_case("neck x4", lambda: [x.clone().requires_grad_()],
             lambda lv_: conv_r(F.interpolate(lv_[0], scale_factor=4, mode="bilinear", align_corners=False)), conv_r, bn_r, build_neck)
    # ---- fpn_bottleneck
    sizes = [144, 72, 36, 18]
    lv = [q(rnd(B, 256, s_, s_, seed=50 + i), dtype) for i, s_ in enumerate(sizes)]
    conv_r, bn_r = nn.Conv2d(1024, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256)
    with torch.no_grad():
        conv_r.weight.copy_(q(rnd(256, 1024, 3, 3, seed=55) * 0.02, dtype))
        bn_r.weight.copy_(rnd(256, seed=56).abs() + 0.5)
        bn_r.bias.copy_(rnd(256, seed=57) * 0.1)

    def build_fpn(conv, norm, gy):
        xs = [t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_() for t in lv]
        with torch.no_grad():
            zs = [ops.conv_gemm(t.detach(), gnn.tap_weight(conv.weight, dtype, 256 * (j + 1), 256 * (j + 2))) for j, t in enumerate(xs[1:])]
            pre = ops.conv_gemm(xs[0].detach(), gnn.slice_weight(conv.weight, dtype, 0, 256), R=3, S=3, pad=1,
                                resid=ops.resize_conv3x3_fwd_sum(zs, (144, 144)))
        y = gnn.concat_resize_conv_bn_act(xs, conv, norm.train(), relu=True)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV, dtype))
        return pre.float().cpu().permute(0, 3, 1, 2), {"out": y.detach().permute(0, 3, 1, 2), "dw": conv.weight.grad,
                                                      "dgamma": norm.weight.grad, "dbeta": norm.bias.grad,
                                                      **{f"d input {i}": xg.grad.permute(0, 3, 1, 2) for i, xg in enumerate(xs)}}

    def fpn_ref(leaves):
        return conv_r(torch.cat([leaves[0]] + [F.interpolate(t, size=(144, 144), mode="bilinear", align_corners=False)
                                               for t in leaves[1:]], 1))

    run_case("fpn_bottleneck", lambda: [t.clone().requires_grad_() for t in lv], fpn_ref, conv_r, bn_r, build_fpn)
    for k, v in res.items():
        print(f"  {str(dtype).split('.')[-1]} production shape, relative L2: {k:52s} {v:.2e}")
    return res


def test_resized_conv_nodes_bf16_at_production_shape():
    """bf16: (a) plain f32 reference: the forward output is held to 1e-2 (measured 3e-3), dgamma to 1e-2, the other gradients
    to 6e-2 -- a 0.3 % difference of the conv output flips the ReLU mask of every pixel whose BatchNorm output is that close
    to zero, and a fraction f of flipped pixels costs sqrt(f) of the gradient's L2 norm (0.1 % -> 3 %; measured 3.2 % on dx,
    dw, dbeta alike, whatever kernel produced them; torch's own bf16 autocast behaves the same); (b) at the build's forward:
    every tensor to 1e-2 (measured <= 3.4e-3: bf16 gradient storage, the nine gathered maps, the low-resolution GEMMs)."""
    res = _resized_conv_nodes_at_production_shape(torch.bfloat16)

    def bound(k):
        if "build's forward" in k:
            return 1e-2
        return 1e-2 if (" out " in k or " dgamma " in k) else 6e-2
    bad = {k: v for k, v in res.items() if not v <= bound(k)}
    assert not bad, bad


def _plain_conv_modules_at_production_shape(dtype):
    """The PLAIN (not restructured) training ConvModules of UperNet at their production shapes, B = 2: `fpn_convs[0]` (3x3, 256 ->
    256 on the 144^2 map: shared-staging implicit GEMM forward, BatchNorm statistics / apply, BatchNorm backward sums + dx, the
    3x3 data gradient and the row-segment weight gradient with its XCD-grouped tiles) and `lateral_convs[0]` (1x1, 768 -> 256 at
    144^2: persistent tile with the statistics-emitting epilogue, 256^2 weight-gradient tile with split-K) -- conv -> BN(batch
    statistics) -> ReLU (models/utils.py:10-52, upernet.py:79-101) against torch f32 autograd on the CPU fed the same operands,
    (a) as a plain f32 graph and (b) evaluated AT the build's own conv output (same ReLU mask and statistics on both sides).
    Returns {"<module> <tensor> [<reference>]": relative L2 error}."""
    import copy
    from torch import nn
    B, res = 2, {}
    for tag, cin, cout, k in (("fpn 3x3 256->256 @144", 256, 256, 3), ("lateral 1x1 768->256 @144", 768, 256, 1)):
        x = q(rnd(B, cin, 144, 144, seed=60 + k), dtype)
        conv_r, bn_r = nn.Conv2d(cin, cout, k, padding=k // 2, bias=False), nn.BatchNorm2d(cout)
        with torch.no_grad():
            conv_r.weight.copy_(q(rnd(cout, cin, k, k, seed=61 + k) * (0.02 if k == 3 else 0.05), dtype))
            bn_r.weight.copy_(rnd(cout, seed=62 + k).abs() + 0.5)
            bn_r.bias.copy_(rnd(cout, seed=63 + k) * 0.1)
        conv, norm = copy.deepcopy(conv_r).to(DEV).to(memory_format=torch.channels_last), copy.deepcopy(bn_r).to(DEV)
        gy = q(rnd(B, cout, 144, 144, seed=64 + k), dtype)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_()
        with torch.no_grad():
            pre = ops.conv_gemm(xd.detach(), gnn.gemm_weight(conv.weight, dtype), R=k, S=k, pad=k // 2)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = gnn.conv_bn_act(xd, conv, norm.train(), relu=True)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV, y.dtype))
        got = {"out": y.detach().permute(0, 3, 1, 2), "d input": xd.grad.permute(0, 3, 1, 2), "dw": conv.weight.grad,
               "dgamma": norm.weight.grad, "dbeta": norm.bias.grad}
        pre_cpu = pre.float().cpu().permute(0, 3, 1, 2)
        for mode in ("f32", "at the build's forward"):
            for p_ in list(conv_r.parameters()) + list(bn_r.parameters()):
                p_.grad = None
            xr = x.clone().requires_grad_()
            yc = conv_r(xr)
            if mode != "f32":
                yc = yc + (pre_cpu - yc).detach()
            yr = F.relu(bn_r.train()(yc))
            yr.backward(gy)
            ref = {"out": yr.detach(), "d input": xr.grad, "dw": conv_r.weight.grad, "dgamma": bn_r.weight.grad, "dbeta": bn_r.bias.grad}
            for name, v in got.items():
                res[f"{tag} {name} [{mode}]"] = _rel_l2(v, ref[name])
    return res


def test_plain_conv_modules_f32_at_production_shape():
    """Round-4 review, parity item 1: the BatchNorm-backward / data-gradient / weight-gradient chain of the 144^2 ConvModules had
    per-op tests at toy sizes only, and the model-level gradient tolerance is 2e-2.  Here the two heaviest plain ConvModules at
    their production shape in f32: within 2e-5 relative L2 of torch autograd evaluated at the build's own forward (same ReLU mask,
    same statistics: what remains is the arithmetic of the backward kernels), within 5e-3 of the plain f32 graph."""
    res = _plain_conv_modules_at_production_shape(torch.float32)
    print({k: f"{v:.2e}" for k, v in res.items()})
    # measured (MI355X, round 5): <= 1.4e-6 at the build's forward, <= 5.1e-4 against the plain graph
    bad = {k: v for k, v in res.items() if not v <= (2e-5 if "build's forward" in k else 5e-3)}
    assert not bad, bad


def test_plain_conv_modules_bf16_at_production_shape():
    """The same in the benchmarked dtype: bf16 operands, f32 accumulation, BatchNorm in f32 on the bf16 conv output.  Against f32
    torch autograd at the build's own forward the remaining error is the bf16 rounding of the stored activations / gradients
    (measured <= 2.4e-3 relative L2 for every tensor; bound 6e-3); against the plain f32 graph measured <= 9.9e-3, bound 2e-2."""
    res = _plain_conv_modules_at_production_shape(torch.bfloat16)
    print({k: f"{v:.2e}" for k, v in res.items()})
    bad = {k: v for k, v in res.items() if not v <= (6e-3 if "build's forward" in k else 2e-2)}
    assert not bad, bad


def test_resized_conv_nodes_f32_at_production_shape():
    """f32 (round-3 review): the end-to-end gradient tolerance of the model tests is 2e-2 for deep layers because f32 round-off
    of the forward flips ReLU masks; this test backs it with a TIGHT one.  The same two nodes in f32 against torch autograd
    evaluated at the build's own forward activations (same masks, same statistics): every gradient and the output within 2e-5
    relative L2 (the review asked for 1e-3) -- an arithmetic error of the low-resolution backward (gather, tap GEMMs, BatchNorm backward) above that would
    show here although the model-level test let it pass.  Against the plain f32 graph (its own masks) the bound is 5e-3."""
    res = _resized_conv_nodes_at_production_shape(torch.float32)
    # measured (MI355X, round 4): <= 1.5e-6 at the build's forward, <= 7.5e-4 against the plain graph
    bad = {k: v for k, v in res.items() if not v <= (2e-5 if "build's forward" in k else 5e-3)}
    assert not bad, bad
