"""Debug (round 6): the failing case of test_resize_conv3x3_fwd_sum_rolling_window, call for call, with the wrong elements located."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
sys.path.insert(0, str(ROOT / "tests"))
from gdlhip import ops  # noqa: E402
from test_hip_ops import rnd, q  # noqa: E402
B, H, W, N, f = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (5, 16, 512, 512, 4)
DEV = torch.device("cuda")
dtype = torch.bfloat16
z = q(rnd(B, H // f, W // f, 9 * N, seed=31 + f), dtype)
add = rnd(N, seed=5)
zd = z.to(DEV, dtype)
y = ops.resize_conv3x3_fwd_sum([zd], (H, W))
y2 = ops.resize_conv3x3_fwd_sum([zd], (H, W), addvec=add.to(DEV), relu=True)
torch.cuda.synchronize()
ref = torch.relu(y.float() + add.to(DEV))
bad = (y2.float() - ref).abs() > 0.05 * ref.abs().max()
print("bad elements", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    for d, name in enumerate(("image", "row", "col", "channel")):
        u = idx[:, d].unique()
        print("  ", name, u[:64].tolist(), "..." if len(u) > 64 else "", len(u))
    for i0 in idx[:6].tolist():
        print("  at", i0, "got", y2[tuple(i0)].item(), "want", ref[tuple(i0)].item(), "plain", y[tuple(i0)].item(), "add", add[i0[3]].item())
import ctypes
from gdlhip import _lib
lib = _lib.load()
lib.gdl_debug_set_tapsum_roll.argtypes = [ctypes.c_int]
y2c = y2.clone()
yc = y.clone()
rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
y3, mean, var = ops.resize_conv3x3_fwd_sum_bn([zd], (H, W), addvec=add.to(DEV), running_mean=rm, running_var=rv, momentum=0.1)
torch.cuda.synchronize()
print("after the statistics call: y2 changed", int((y2 != y2c).sum()), "y changed", int((y != yc).sum()), "y3 vs y+add bad",
      int(((y3.float() - (y.float() + add.to(DEV))).abs() > 0.3).sum()))
lib.gdl_debug_set_tapsum_roll(0)
v = ops.resize_conv3x3_fwd_sum([zd], (H, W))
torch.cuda.synchronize()
print("after version 2 plain: y2 changed", int((y2 != y2c).sum()), "y changed", int((y != yc).sum()))
v3, vmean, vvar = ops.resize_conv3x3_fwd_sum_bn([zd], (H, W), addvec=add.to(DEV))
torch.cuda.synchronize()
print("after version 2 statistics: y2 changed", int((y2 != y2c).sum()), "y changed", int((y != yc).sum()))
lib.gdl_debug_set_tapsum_roll(1)
bad = y2 != y2c
if bad.any():
    idx = bad.nonzero()
    for d, name in enumerate(("image", "row", "col", "channel")):
        u = idx[:, d].unique()
        print("  ", name, u[:64].tolist(), "..." if len(u) > 64 else "", len(u))
