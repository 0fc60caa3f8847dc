"""CPU checks (pure torch, f64) of the algebraic restructurings the MI355X path uses -- the GPU tests hold the KERNELS to torch
autograd, these hold the IDENTITIES themselves, independent of any kernel:

* backward of conv3x3(pad 1)(bilinear resize(x)) through nine low-resolution maps G_t = resize^T shift_t^T dy
  (gdlhip/ops.py:resize_conv3x3_bwd; reference layer: multilevel_neck.py:56-67,157-158, upernet.py:144-152);
* forward of the same layers through nine low-resolution tap products: conv3x3(resize(x)) = sum_t shift_t(resize(W_t x))
  (gdlhip/ops.py:resize_conv3x3_fwd_sum), also over a concat of upsampled levels (upernet.py:144-152);
* SegFormer's linear_fuse: a 1x1 convolution commutes with the bilinear resize (gdlhip/nn.py:pyramid_fuse_bn_act;
  segformer_mlp.py:97-125);
* the image stems (7x7 / 2 of torchvision's ResNet, 7x7 / 4 of MiT) as 3x3 sub-pixel-phase convolutions on the
  space-to-depth image (gdlhip/cnn.py:mark_stem) -- the product's own index tables drive a plain-torch evaluation.
"""

import pytest
import torch
import torch.nn.functional as F


def _gathered_maps(dy, hi, wi):
    """G[t][b, n, q] = d/dx[b, n, q] of <dy, shift_t(resize(x))>: one map per filter tap t = 3 r + s (offset (r-1, s-1))."""
    B, N, Ho, Wo = dy.shape
    x = torch.zeros(B, N, hi, wi, dtype=dy.dtype, requires_grad=True)
    up = F.pad(F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False), (1, 1, 1, 1))
    maps = []
    for r in range(3):
        for s in range(3):
            (g,) = torch.autograd.grad((up[:, :, r:r + Ho, s:s + Wo] * dy).sum(), x, retain_graph=True)
            maps.append(g)
    return maps


@pytest.mark.parametrize("factor", [2, 4, 8])
def test_resized_conv_gradients_from_nine_low_resolution_maps(factor):
    torch.manual_seed(factor)
    B, C, N, hi, wi = 2, 5, 7, 3, 4
    x = torch.randn(B, C, hi, wi, dtype=torch.float64, requires_grad=True)
    w = torch.randn(N, C, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(F.interpolate(x, scale_factor=factor, mode="bilinear", align_corners=False), w, padding=1)
    dy = torch.randn_like(y)
    dx_ref, dw_ref = torch.autograd.grad((y * dy).sum(), (x, w))
    G = _gathered_maps(dy, hi, wi)                                    # nine [B, N, hi, wi] maps
    # dx = sum_t W_t^T G_t        (a GEMM over the LOW-resolution pixels with K = 9 N)
    dx = sum(torch.einsum("nc,bnhw->bchw", w[:, :, t // 3, t % 3], G[t]) for t in range(9))
    # dW_t = sum_q G_t[q] (x) x[q]  (a weight gradient over the low-resolution pixels)
    dw = torch.stack([torch.einsum("bnhw,bchw->nc", G[t], x) for t in range(9)], -1).view(N, C, 3, 3)
    assert torch.allclose(dx, dx_ref, rtol=1e-10, atol=1e-12)
    assert torch.allclose(dw, dw_ref, rtol=1e-10, atol=1e-12)


def test_one_by_one_convolution_commutes_with_bilinear_resize():
    torch.manual_seed(0)
    B, E, N = 2, 6, 5
    sizes = [(2, 3), (4, 6), (8, 12), (16, 24)]
    levels = [torch.randn(B, E, h, w, dtype=torch.float64) for h, w in sizes]
    w = torch.randn(N, 4 * E, 1, 1, dtype=torch.float64)
    cat = torch.cat([F.interpolate(l, size=sizes[-1], mode="bilinear", align_corners=False) for l in levels[:-1]] + [levels[-1]], 1)
    ref = F.conv2d(cat, w)
    per_level = F.conv2d(levels[-1], w[:, 3 * E:])
    for j, l in enumerate(levels[:-1]):
        per_level = per_level + F.interpolate(F.conv2d(l, w[:, j * E:(j + 1) * E]), size=sizes[-1], mode="bilinear",
                                              align_corners=False)
    assert torch.allclose(per_level, ref, rtol=1e-10, atol=1e-12)


def tap_sum(z, size):
    """sum_t shift_t(resize(z_t)): z [B, 9, N, h, w] tap products at low resolution -> [B, N, H, W]; shift_t reads position
    p + (r - 1, s - 1) of the resized map, zero outside (= the convolution's zero padding).  Also the reference of the GPU test."""
    B, _, N, _, _ = z.shape
    H, W = size
    out = z.new_zeros(B, N, H, W)
    for t in range(9):
        up = F.pad(F.interpolate(z[:, t], size=size, mode="bilinear", align_corners=False), (1, 1, 1, 1))
        out = out + up[:, :, t // 3:t // 3 + H, t % 3:t % 3 + W]
    return out


@pytest.mark.parametrize("factor", [2, 4, 8])
def test_resized_conv_forward_from_nine_low_resolution_tap_products(factor):
    torch.manual_seed(factor)
    B, C, N, hi, wi = 2, 5, 7, 3, 4
    x = torch.randn(B, C, hi, wi, dtype=torch.float64)
    w = torch.randn(N, C, 3, 3, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=factor, mode="bilinear", align_corners=False), w, padding=1)
    z = torch.stack([torch.einsum("nc,bchw->bnhw", w[:, :, t // 3, t % 3], x) for t in range(9)], 1)   # 1x1 convs, 9 N outputs
    assert torch.allclose(tap_sum(z, (factor * hi, factor * wi)), ref, rtol=1e-10, atol=1e-12)


def test_conv_over_concat_of_upsampled_levels_per_level():
    torch.manual_seed(1)
    B, C, N = 2, 4, 6
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3)]
    levels = [torch.randn(B, C, h, w, dtype=torch.float64) for h, w in sizes]
    w = torch.randn(N, 4 * C, 3, 3, dtype=torch.float64)
    cat = torch.cat([levels[0]] + [F.interpolate(l, size=sizes[0], mode="bilinear", align_corners=False) for l in levels[1:]], 1)
    ref = F.conv2d(cat, w, padding=1)
    got = F.conv2d(levels[0], w[:, :C], padding=1)
    for j in (1, 2, 3):
        wj = w[:, j * C:(j + 1) * C]
        z = torch.stack([torch.einsum("nc,bchw->bnhw", wj[:, :, t // 3, t % 3], levels[j]) for t in range(9)], 1)
        got = got + tap_sum(z, sizes[0])
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("C,k,s,pad", [(3, 7, 2, 3), (3, 7, 4, 3), (5, 7, 2, 3), (4, 3, 2, 1), (10, 7, 4, 3)])
def test_strided_stem_as_phase_convolutions_on_the_space_to_depth_image(C, k, s, pad):
    """conv k x k / stride s on raw bands == (4 / s)^2 phase convolutions 3x3 / stride 1 on the 4 x 4 space-to-depth image, each
    writing the output pixels [ey::e, ex::e]; and the parameter gradient == the phase filters' gradients scattered back to the
    k x k taps.  Uses gdlhip.cnn's index tables (pure torch, CPU) with F.conv2d standing in for the kernels."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "geo-deep-learning_amd"))
    from gdlhip import cnn
    torch.manual_seed(C * k + s)
    N, H = 6, 32
    w = torch.nn.Parameter(torch.randn(N, C, k, k, dtype=torch.float64))
    x = torch.randn(2, C, H, H, dtype=torch.float64)
    dy = torch.randn(2, N, H // s, H // s, dtype=torch.float64)
    ref = F.conv2d(x, w, stride=s, padding=pad)
    (dw_ref,) = torch.autograd.grad((ref * dy).sum(), w)
    cnn.mark_stem(w, s, pad)
    e, idx, valid = cnn._stem_index(w)
    assert e == 4 // s and cnn._wshape(w) == (N, 16 * C, 3, 3)
    flat = w.detach().reshape(N, -1)
    filt = torch.where(valid.unsqueeze(1), flat[:, idx].permute(1, 0, 2, 3), flat.new_zeros(()))      # [phase, N, 9, 16 C]
    xs = F.pixel_unshuffle(x, 4)                                                                       # channel order (c, dy, dx)
    out = torch.zeros_like(ref)
    dw = torch.zeros(N, C * k * k, dtype=torch.float64)
    for ph in range(e * e):
        wd = filt[ph].view(N, 3, 3, -1).permute(0, 3, 1, 2).clone().requires_grad_(True)
        y = F.conv2d(xs, wd, padding=1)
        out[:, :, ph // e::e, ph % e::e] = y.detach()
        (g,) = torch.autograd.grad((y * dy[:, :, ph // e::e, ph % e::e]).sum(), wd)
        dw[:, idx[ph][valid[ph]]] += g.permute(0, 2, 3, 1).reshape(N, 9, -1)[:, valid[ph]]
    assert torch.allclose(out, ref, rtol=1e-10, atol=1e-12)
    assert torch.allclose(dw.view_as(dw_ref), dw_ref, rtol=1e-10, atol=1e-12)
    # every filter tap is used exactly once per phase
    for ph in range(e * e):
        assert sorted(idx[ph][valid[ph]].tolist()) == list(range(C * k * k))
