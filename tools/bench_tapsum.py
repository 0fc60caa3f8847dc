#!/usr/bin/env python
"""GPU micro-benchmark of the low-resolution FORWARD of conv3x3(bilinear resize(x)) at DOFA-base + UperNet's batch-32 shapes
(neck x4 / x2 levels 768 -> 768, UperNet fpn_bottleneck over [144^2, 72^2, 36^2, 18^2] x 256): tap products (one 1x1 GEMM at
low resolution) + gather-sum (gdl_resize_conv3x3_fwd_sum, 8- and 16-byte vectors) against the round-2 forward (sub-pixel
phases for x4, upsample -> 3x3 for x2, concat buffer -> 3x3 for the bottleneck), interleaved in one process."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402
from gdlhip import nn as gnn  # noqa: E402

lib = _lib.load()
lib.gdl_debug_set_tapsum_vec.argtypes = [ctypes.c_int]
lib.gdl_debug_set_tapsum_mfma.argtypes = [ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bf = torch.bfloat16


def timeit(fn, rounds=5, inner=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def conv_module(c, n, bias):
    conv = torch.nn.Conv2d(c, n, 3, padding=1, bias=bias).cuda().to(memory_format=torch.channels_last)
    norm = torch.nn.BatchNorm2d(n).cuda().eval()
    return conv, norm


with torch.no_grad():
    for up in (4, 2):
        conv, norm = conv_module(768, 768, True)
        x = torch.randn(B, 36, 36, 768, device="cuda").to(bf)
        wt = gnn.tap_weight(conv.weight, bf)
        z = ops.conv_gemm(x, wt)
        size = (36 * up, 36 * up)
        t_gemm = timeit(lambda: ops.conv_gemm(x, wt))
        res = []
        lib.gdl_debug_set_tapsum_mfma(0)
        for vec in (8, 4):
            lib.gdl_debug_set_tapsum_vec(vec)
            res.append(timeit(lambda: ops.resize_conv3x3_fwd_sum([z], size)))
        lib.gdl_debug_set_tapsum_vec(0)
        ref = ops.resize_conv3x3_fwd_sum([z], size)
        for mode in (5, 2, 4):
            lib.gdl_debug_set_tapsum_mfma(mode)
            res.append(timeit(lambda: ops.resize_conv3x3_fwd_sum([z], size)))
            dev = (ops.resize_conv3x3_fwd_sum([z], size).float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
            res.append(dev)
        lib.gdl_debug_set_tapsum_mfma(1)
        new = timeit(lambda: gnn.conv_bn_act(x, conv, norm, relu=True, up=up))
        gnn.FUSE_TAPSUM = False
        old = timeit(lambda: gnn.conv_bn_act(x, conv, norm, relu=True, up=up))
        gnn.FUSE_TAPSUM = True
        out_b = B * size[0] * size[1] * 768 * 2
        print(f"neck x{up} 768->768 @36^2 -> {size[0]}^2: tap GEMM {t_gemm:6.0f} us ({2 * B * 1296 * 768 * 6912 / t_gemm / 1e6:5.0f} TF/s), "
              f"gather-sum vec8 {res[0]:6.0f} us / vec4 {res[1]:6.0f} us / MFMA v2 {res[2]:6.0f} us (dev {res[3]:.1e}) / v1 32-col {res[4]:6.0f} us (dev {res[5]:.1e}) / v1 16-col {res[6]:6.0f} us "
              f"({(out_b + z.numel() * 2) / min(res[0], res[1], res[2], res[4], res[6]) / 1e3:5.0f} GB/s), "
              f"node (eval) new {new:6.0f} us vs round-2 {old:6.0f} us", flush=True)

    conv, norm = conv_module(1024, 256, False)
    lv = [torch.randn(B, s, s, 256, device="cuda").to(bf) for s in (144, 72, 36, 18)]
    zs = [ops.conv_gemm(l, gnn.tap_weight(conv.weight, bf, 256 * (j + 1), 256 * (j + 2))) for j, l in enumerate(lv[1:])]
    t_gemm = [timeit(lambda l=l, j=j: ops.conv_gemm(l, gnn.tap_weight(conv.weight, bf, 256 * (j + 1), 256 * (j + 2)))) for j, l in enumerate(lv[1:])]
    res = []
    lib.gdl_debug_set_tapsum_mfma(0)
    for vec in (8, 4):
        lib.gdl_debug_set_tapsum_vec(vec)
        res.append(timeit(lambda: ops.resize_conv3x3_fwd_sum(zs, (144, 144))))
    lib.gdl_debug_set_tapsum_vec(0)
    for mode in (5, 2):
        lib.gdl_debug_set_tapsum_mfma(mode)
        res.append(timeit(lambda: ops.resize_conv3x3_fwd_sum(zs, (144, 144))))
    lib.gdl_debug_set_tapsum_mfma(1)
    w0 = gnn.slice_weight(conv.weight, bf, 0, 256)
    r = ops.resize_conv3x3_fwd_sum(zs, (144, 144))
    t0 = timeit(lambda: ops.conv_gemm(lv[0], w0, R=3, S=3, pad=1, resid=r))
    new = timeit(lambda: gnn.concat_resize_conv_bn_act(lv, conv, norm, relu=True))
    gnn.FUSE_TAPSUM = False
    old = timeit(lambda: gnn.concat_resize_conv_bn_act(lv, conv, norm, relu=True))
    gnn.FUSE_TAPSUM = True
    print(f"fpn_bottleneck 4x256 -> 256 @144^2: tap GEMMs {t_gemm[0]:5.0f} + {t_gemm[1]:5.0f} + {t_gemm[2]:5.0f} us, gather-sum (3 sources) "
          f"vec8 {res[0]:6.0f} us / vec4 {res[1]:6.0f} us / MFMA v2 {res[2]:6.0f} us / v1 {res[3]:6.0f} us, native 3x3 + residual {t0:6.0f} us, node (eval) new {new:6.0f} us vs "
          f"round-2 {old:6.0f} us", flush=True)
