"""GPU parity tests for the transformer-block backward kernels and the block-level autograd nodes:
each HIP path (through the C-ABI) vs torch autograd of a plain f32 PyTorch statement of the same
op on seeded inputs.  f32 tolerance 2e-4 of the reference's scale, bf16 3e-2."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

gdlhip = pytest.importorskip("gdlhip")
from gdlhip import ops, tnn  # noqa: E402

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 2e-4 if dtype == torch.float32 else 3e-2


def close(got, ref, dtype, what="", factor=1.0):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    s = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= factor * tol(dtype) * s, f"{what}: max err {err:.3e} vs scale {s:.3e}"


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed + 7 * len(shape) + sum(shape))
    return torch.randn(*shape, generator=g) * std


def q(t, dtype):
    return t.to(dtype).float()


@pytest.mark.parametrize("dy_dtype", DTYPES)
@pytest.mark.parametrize("rows,D", [(37, 64), (300, 320), (1297, 768), (50, 1024), (1003, 32), (515, 128), (77, 96), (6, 64)])
def test_layernorm_bwd(dy_dtype, rows, D):
    x = rnd(rows, D).requires_grad_()
    gamma, beta = (1 + 0.2 * rnd(D, seed=1)).requires_grad_(), rnd(D, seed=2).requires_grad_()
    dy, dres = q(rnd(rows, D, seed=3), dy_dtype), rnd(rows, D, seed=4)
    y = F.layer_norm(x, (D,), gamma, beta, 1e-6)
    y.backward(dy)
    dx, dg, db = ops.layernorm_bwd(x.detach().to(DEV), dy.to(DEV, dy_dtype), gamma.detach().to(DEV), 1e-6,
                                   dres=dres.to(DEV))
    close(dx, x.grad + dres, torch.float32, "ln dx")
    close(dg, gamma.grad, torch.float32, "ln dgamma", 3)
    close(db, beta.grad, torch.float32, "ln dbeta", 3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum_and_layerscale_bwd(dtype):
    B, N, C = 3, 150, 320
    x = q(rnd(B, N, C), dtype)
    close(ops.colsum(x.to(DEV, dtype)), x.sum((0, 1)), torch.float32, "colsum", 5)
    g, z = rnd(B, N, C, seed=1), q(rnd(B, N, C, seed=2), dtype)
    gamma, s = 0.1 * rnd(C, seed=3), torch.tensor([0.0, 1.25, 1.25])
    dz, dg = ops.layerscale_bwd(g.to(DEV), z.to(DEV, dtype), gamma.to(DEV), s.to(DEV), dtype)
    close(dz, g * s.view(B, 1, 1) * gamma, dtype, "layerscale dz")
    close(dg, (g * s.view(B, 1, 1) * z).sum((0, 1)), torch.float32, "layerscale dgamma", 5)
    dz, none = ops.layerscale_bwd(g.to(DEV), None, None, None, dtype)
    assert none is None
    close(dz, g, dtype, "plain cast")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 256), (1, 9, 7, 64), (2, 5, 70, 128), (1, 1, 1, 64), (2, 3, 33, 320)])
def test_dwconv_gelu_bwd(dtype, B, H, W, C):
    u = q(rnd(B, H, W, C), dtype).requires_grad_()
    w = (0.3 * rnd(C, 1, 3, 3, seed=1)).requires_grad_()
    bias = (0.1 * rnd(C, seed=2)).requires_grad_()
    dy = q(rnd(B, H, W, C, seed=3), dtype)
    y = F.gelu(F.conv2d(u.permute(0, 3, 1, 2), w, bias, padding=1, groups=C)).permute(0, 2, 3, 1)
    y.backward(dy)
    w9 = w.detach().reshape(C, 9).t().contiguous()
    du, dw9, db = ops.dwconv3x3_gelu_bwd(u.detach().to(DEV, dtype), dy.to(DEV, dtype), w9.to(DEV), bias.detach().to(DEV))
    close(du, u.grad, dtype, "dw du")
    close(dw9.t().reshape(C, 1, 3, 3), w.grad, dtype, "dw dw", 2)
    close(db, bias.grad, dtype, "dw db", 2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Nq,Nkv,H,hd", [(2, 70, 70, 2, 64), (1, 256, 64, 1, 64), (2, 197, 50, 5, 64), (2, 100, 36, 5, 32)])
def test_attention_bwd(dtype, B, Nq, Nkv, H, hd):
    D = H * hd
    qf = q(rnd(B, Nq, D), dtype).requires_grad_()
    kvf = q(rnd(B, Nkv, 2 * D, seed=1), dtype).requires_grad_()
    do = q(rnd(B, Nq, D, seed=2), dtype)

    def heads(t, n):
        return t.reshape(B, n, H, hd).transpose(1, 2)
    kf, vf = kvf[..., :D], kvf[..., D:]
    att = (heads(qf, Nq) @ heads(kf, Nkv).transpose(-1, -2) * hd ** -0.5).softmax(-1)
    out = (att @ heads(vf, Nkv)).transpose(1, 2).reshape(B, Nq, D)
    out.backward(do)
    qd, kvd = qf.detach().to(DEV, dtype), kvf.detach().to(DEV, dtype)
    dq = torch.empty_like(qd)
    dkv = torch.empty_like(kvd)
    ops.attention_bwd(qd, kvd[..., :D], kvd[..., D:], do.to(DEV, dtype), H, dq, dkv[..., :D], dkv[..., D:])
    close(dq, qf.grad, dtype, "dq")
    close(dkv, kvf.grad, dtype, "dkv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,s,p,C,N,hw", [(3, 2, 1, 64, 128, 16), (8, 8, 0, 64, 64, 32), (2, 2, 0, 320, 320, 8),
                                          (3, 1, 1, 64, 64, 9), (1, 1, 0, 128, 64, 5),
                                          # MiT-B0 widths: channel tails in bf16
                                          (8, 8, 0, 32, 32, 32), (3, 2, 1, 32, 64, 16), (3, 2, 1, 160, 256, 8),
                                          (1, 1, 0, 160, 32, 7)])
def test_conv_node(dtype, k, s, p, C, N, hw):
    B = 2
    x = q(rnd(B, hw, hw, C), dtype).requires_grad_()
    conv = torch.nn.Conv2d(C, N, k, s, p).to(memory_format=torch.channels_last)
    with torch.no_grad():
        conv.weight.copy_(q(conv.weight, dtype))
    y = conv(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    dy = q(rnd(*y.shape, seed=5), dtype)
    y.backward(dy)
    ref = (x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    cg = torch.nn.Conv2d(C, N, k, s, p).to(DEV).to(memory_format=torch.channels_last)
    cg.load_state_dict(conv.state_dict())
    xg = x.detach().to(DEV, dtype).requires_grad_()
    yg = tnn.conv(xg, cg.weight, cg.bias, stride=s, pad=p, out_dtype=torch.float32)
    close(yg, y, dtype, "conv fwd")
    yg.backward(dy.to(DEV))
    close(xg.grad, ref[0], dtype, "conv dx")
    close(cg.weight.grad, ref[1], dtype, "conv dw", 2)
    close(cg.bias.grad, ref[2], dtype, "conv db", 2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gelu_grad_epilogue_and_aux_out(dtype):
    B, N, K, Hd = 2, 150, 128, 256
    x, w, b = q(rnd(B, N, K), dtype), q(0.1 * rnd(Hd, K, seed=1), dtype), 0.1 * rnd(Hd, seed=2)
    u_ref = x @ w.t() + b
    u = torch.empty((B, N, Hd), device=DEV, dtype=dtype)
    f = ops.linear(x.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV), act=ops.ACT_GELU, aux_out=u)
    close(u, u_ref, dtype, "aux_out")
    close(f, F.gelu(u_ref), dtype, "gelu out")
    dz, w2 = q(rnd(B, N, K, seed=3), dtype), q(0.1 * rnd(K, Hd, seed=4), dtype)
    uq = u.float().cpu().requires_grad_()
    (F.gelu(uq) @ w2.t()).backward(dz)
    w2t = w2.t().contiguous()                      # [Hd, K]: the "transposed" operand
    du = ops.linear(dz.to(DEV, dtype), w2t.to(DEV, dtype), None, resid=u, act=ops.ACT_MUL_GELU_GRAD)
    close(du, uq.grad, dtype, "gelu-grad epilogue")


def _mit_ref(x, prm, s1, s2, hh, ww, heads, sr, eps):
    n1w, n1b, qw, qb, kvw, kvb, pw, pb, n2w, n2b, f1w, f1b, dww, dwb, f2w, f2b = prm[:16]
    B, N, C = x.shape
    hd = C // heads
    h1 = F.layer_norm(x, (C,), n1w, n1b, eps)
    qq = F.linear(h1, qw, qb).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        srw, srb, nsw, nsb = prm[16:20]
        red = F.conv2d(h1.permute(0, 2, 1).reshape(B, C, hh, ww), srw, srb, stride=sr)
        xn = F.layer_norm(red.reshape(B, C, -1).permute(0, 2, 1), (C,), nsw, nsb, 1e-5)
    else:
        xn = h1
    kv = F.linear(xn, kvw, kvb).reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    att = (qq @ kv[0].transpose(-2, -1) * hd ** -0.5).softmax(-1)
    a = (att @ kv[1]).transpose(1, 2).reshape(B, N, C)
    x1 = x + s1.view(B, 1, 1) * F.linear(a, pw, pb)
    h2 = F.layer_norm(x1, (C,), n2w, n2b, eps)
    u = F.linear(h2, f1w, f1b)
    g = F.gelu(F.conv2d(u.transpose(1, 2).reshape(B, -1, hh, ww), dww, dwb, padding=1, groups=u.shape[-1]))
    return x1 + s2.view(B, 1, 1) * F.linear(g.flatten(2).transpose(1, 2), f2w, f2b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,heads,sr,hh", [(64, 1, 4, 16), (128, 2, 1, 8), (320, 5, 2, 8)])
def test_mit_block_node(dtype, C, heads, sr, hh):
    B, ww = 2, hh
    N = hh * ww
    hid = 4 * C

    def lin(o, i, seed):
        return rnd(o, i, seed=seed, std=(1.0 / i) ** 0.5), 0.05 * rnd(o, seed=seed + 1)
    n1w, n1b = 1 + 0.1 * rnd(C, seed=10), 0.1 * rnd(C, seed=11)
    qw, qb = lin(C, C, 12)
    kvw, kvb = lin(2 * C, C, 14)
    pw, pb = lin(C, C, 16)
    n2w, n2b = 1 + 0.1 * rnd(C, seed=18), 0.1 * rnd(C, seed=19)
    f1w, f1b = lin(hid, C, 20)
    dww, dwb = 0.3 * rnd(hid, 1, 3, 3, seed=22), 0.05 * rnd(hid, seed=23)
    f2w, f2b = lin(C, hid, 24)
    prm = [n1w, n1b, qw, qb, kvw, kvb, pw, pb, n2w, n2b, f1w, f1b, dww, dwb, f2w, f2b]
    if sr > 1:
        srw = rnd(C, C, sr, sr, seed=26, std=(1.0 / (C * sr * sr)) ** 0.5)
        prm += [srw, 0.05 * rnd(C, seed=27), 1 + 0.1 * rnd(C, seed=28), 0.1 * rnd(C, seed=29)]
    x = rnd(B, N, C, seed=1)
    s1, s2 = torch.tensor([1.0, 1.25]), torch.tensor([1.25, 0.0])
    gout = rnd(B, N, C, seed=2)

    ref_p = [p.clone().requires_grad_() for p in prm]
    xr = x.clone().requires_grad_()
    yr = _mit_ref(xr, ref_p, s1, s2, hh, ww, heads, sr, 1e-6)
    yr.backward(gout)

    dev_p = []
    for p in prm:
        t = p.to(DEV)
        if t.dim() == 4 and t.shape[1] > 1:
            t = t.contiguous(memory_format=torch.channels_last)
        dev_p.append(t.requires_grad_())
    xd = x.to(DEV).requires_grad_()
    yd = tnn.mit_block(xd, s1.to(DEV), s2.to(DEV), hh, ww, heads, sr, 1e-6, 1e-5, dtype, tuple(dev_p))
    f = 1.0 if dtype == torch.float32 else 1.5
    close(yd, yr, dtype, "mit fwd", f)
    yd.backward(gout.to(DEV))
    close(xd.grad, xr.grad, dtype, "mit dx", 2 * f)
    names = ["n1w", "n1b", "qw", "qb", "kvw", "kvb", "pw", "pb", "n2w", "n2b", "f1w", "f1b", "dww", "dwb", "f2w", "f2b",
             "srw", "srb", "nsw", "nsb"]
    for nm, pd, pr in zip(names, dev_p, ref_p):
        assert pd.grad is not None, nm
        close(pd.grad, pr.grad, dtype, f"mit d{nm}", 3 * f)


def _vit_ref(x, prm, s1, s2, heads, eps):
    n1w, n1b, qkvw, qkvb, pw, pb, g1, n2w, n2b, f1w, f1b, f2w, f2b, g2 = prm
    B, N, C = x.shape
    hd = C // heads
    h1 = F.layer_norm(x, (C,), n1w, n1b, eps)
    qkv = F.linear(h1, qkvw, qkvb).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = (qkv[0] @ qkv[1].transpose(-2, -1) * hd ** -0.5).softmax(-1)
    a = (att @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    x1 = x + s1.view(B, 1, 1) * (g1 * F.linear(a, pw, pb))
    h2 = F.layer_norm(x1, (C,), n2w, n2b, eps)
    return x1 + s2.view(B, 1, 1) * (g2 * F.linear(F.gelu(F.linear(h2, f1w, f1b)), f2w, f2b))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,heads,N", [(128, 2, 70), (192, 3, 197)])
def test_vit_block_node(dtype, C, heads, N):
    B, hid = 2, 4 * C

    def lin(o, i, seed):
        return rnd(o, i, seed=seed, std=(1.0 / i) ** 0.5), 0.05 * rnd(o, seed=seed + 1)
    qkvw, qkvb = lin(3 * C, C, 12)
    pw, pb = lin(C, C, 16)
    f1w, f1b = lin(hid, C, 20)
    f2w, f2b = lin(C, hid, 24)
    prm = [1 + 0.1 * rnd(C, seed=10), 0.1 * rnd(C, seed=11), qkvw, qkvb, pw, pb, 0.3 * rnd(C, seed=30),
           1 + 0.1 * rnd(C, seed=18), 0.1 * rnd(C, seed=19), f1w, f1b, f2w, f2b, 0.3 * rnd(C, seed=31)]
    x = rnd(B, N, C, seed=1)
    s1, s2 = torch.tensor([1.25, 1.0]), torch.tensor([0.0, 1.25])
    gout = rnd(B, N, C, seed=2)
    ref_p = [p.clone().requires_grad_() for p in prm]
    xr = x.clone().requires_grad_()
    yr = _vit_ref(xr, ref_p, s1, s2, heads, 1e-6)
    yr.backward(gout)
    dev_p = [p.to(DEV).requires_grad_() for p in prm]
    xd = x.to(DEV).requires_grad_()
    yd = tnn.vit_block(xd, s1.to(DEV), s2.to(DEV), heads, 1e-6, dtype, tuple(dev_p))
    f = 1.0 if dtype == torch.float32 else 1.5
    close(yd, yr, dtype, "vit fwd", f)
    yd.backward(gout.to(DEV))
    close(xd.grad, xr.grad, dtype, "vit dx", 2 * f)
    names = ["n1w", "n1b", "qkvw", "qkvb", "pw", "pb", "g1", "n2w", "n2b", "f1w", "f1b", "f2w", "f2b", "g2"]
    for nm, pd, pr in zip(names, dev_p, ref_p):
        assert pd.grad is not None, nm
        close(pd.grad, pr.grad, dtype, f"vit d{nm}", 3 * f)
