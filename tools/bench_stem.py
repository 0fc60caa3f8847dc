#!/usr/bin/env python
"""Image stem 7x7 / stride 2 on raw bands (UNet++ / ResNet, batch 32, 512^2, bf16): the round-1/2 path (strided patchify = an
im2col matrix of 49/4 x the image + GEMM, K = 147 -> 192) against the space-to-depth form (re-layout of the image into 2 x 2
pixel blocks of 4 x 4 = 48 -> 64 channels, then four sub-pixel-phase 3x3 implicit-GEMM convolutions; gdlhip.cnn.mark_stem), forward
and weight gradient."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import cnn, ops  # noqa: E402

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


img = torch.randn(B, 3, 512, 512, device="cuda")
w = torch.nn.Parameter(torch.randn(64, 3, 7, 7, device="cuda") * 0.05)
# old: patchify(k 7, stride 2, pad 3) -> cols [B*256*256, 192], GEMM with the flat filter
wflat = torch.zeros(64, 192, device="cuda")
wflat[:, :147] = w.detach().reshape(64, -1)
wflat = wflat.to(bf)
t_cols = timeit(lambda: ops.patchify(img, 7, 3, 256, 256, 192, bf, stride=2))
cols = ops.patchify(img, 7, 3, 256, 256, 192, bf, stride=2).view(B, 256, 256, 192)
t_gemm = timeit(lambda: ops.conv_gemm(cols, wflat))
y_old = ops.conv_gemm(cols, wflat)
dy = torch.randn_like(y_old)
t_wg_old = timeit(lambda: ops.conv_wgrad(cols, dy, R=1, S=1))
# new: 4 x 4 space-to-depth re-layout + four sub-pixel-phase 3x3 convolutions (gdlhip.cnn.mark_stem)
cnn.mark_stem(w, 2, 3)
t_s2d = timeit(lambda: cnn.space_to_depth_image(img, bf))
xs = cnn.space_to_depth_image(img, bf)
t_conv = timeit(lambda: cnn.stem_conv(xs, w, 64))
y_new = cnn.stem_conv(xs, w, 64)
t_wg_new = timeit(lambda: cnn.stem_param_grad(xs, dy, w))
dev = (y_new.float() - y_old.float()).abs().max().item() / y_old.float().abs().max().item()
dw_old = ops.conv_wgrad(cols, dy, R=1, S=1)[:, :147].reshape(64, 3, 7, 7)
dw_new = cnn.stem_param_grad(xs, dy, w)
dwdev = (dw_new - dw_old).abs().max().item() / dw_old.abs().max().item()
print(f"stem 7x7/2, batch {B}: im2col path: patchify {t_cols:.0f} us + GEMM {t_gemm:.0f} us, weight gradient {t_wg_old:.0f} us | "
      f"space-to-depth path: re-layout {t_s2d:.0f} us + 4 phase convs {t_conv:.0f} us, weight gradient {t_wg_new:.0f} us | "
      f"outputs differ by {dev:.1e}, weight gradients by {dwdev:.1e} (bf16)")
