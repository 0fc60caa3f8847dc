"""debug: batched grouped conv (forward / data gradient / weight gradient) vs torch, bf16 and f32"""
import sys
from pathlib import Path
import torch
import torch.nn.functional as F
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import cnn, ops  # noqa: E402

torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for C, g in ((256, 32), (512, 32), (1024, 32), (2048, 32)):
        b, h, w_ = 2, 20, 24
        wt = (torch.randn(C, C // g, 3, 3) * 0.2).to(dtype).float()
        x = torch.randn(b, C, h, w_).to(dtype).float()
        gy = torch.randn(b, C, h, w_).to(dtype).float()
        xr = x.clone().requires_grad_(True)
        wr = wt.clone().requires_grad_(True)
        y = F.conv2d(xr, wr, padding=1, groups=g)
        y.backward(gy)
        wp = torch.nn.Parameter(wt.cuda())
        cnn.mark_groups(wp, g)
        sg = cnn.supergroups(wp, C, C, dtype)
        fwd, dgr = cnn.grouped_operands(wp, dtype, sg)
        xg = x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
        gg = gy.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
        yo = ops.conv_gemm_grouped(xg, fwd, R=3, S=3, pad=1)
        dx = ops.conv_gemm_grouped(gg, dgr, R=3, S=3, pad=1)
        dw = cnn._grouped_param_grad(ops.conv_wgrad_grouped(xg, gg, Z=sg[0], R=3, S=3, pad=1), wp, sg)
        e = lambda a_, r_: ((a_.float().cpu() - r_).abs().max() / r_.abs().max()).item()  # noqa: E731
        print(dtype, C, g, sg, "fwd %.4f dx %.4f dw %.4f" % (e(yo.permute(0, 3, 1, 2), y.detach()), e(dx.permute(0, 3, 1, 2), xr.grad),
                                                             e(dw, wr.grad)), flush=True)
