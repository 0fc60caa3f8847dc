"""Debug (round 6): repeat the rolling-window gather-sum on one case and count results that differ from the first / from version 2."""
import ctypes
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402
lib = _lib.load()
lib.gdl_debug_set_tapsum_roll.argtypes = [ctypes.c_int]
B, H, W, N = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 5 else (5, 16, 512, 512)
fs = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [4]     # e.g. 4 or 2,4,8
REP = int(sys.argv[6]) if len(sys.argv) > 6 else 40
torch.manual_seed(1)
zs = [torch.randn(B, H // f, W // f, 9 * N, device="cuda").to(torch.bfloat16) for f in fs]
add = torch.randn(N, device="cuda")
lib.gdl_debug_set_tapsum_roll(0)
ref = ops.resize_conv3x3_fwd_sum(zs, (H, W), addvec=add, relu=True).float()
refp = ops.resize_conv3x3_fwd_sum(zs, (H, W)).float()
for mode in (1, 2):
    lib.gdl_debug_set_tapsum_roll(mode)
    for name, kw, r in (("plain", {}, refp), ("addend + relu", dict(addvec=add, relu=True), ref)):
        nbad, runs_bad, where = 0, 0, None
        for it in range(REP):
            y = ops.resize_conv3x3_fwd_sum(zs, (H, W), **kw).float()
            bad = (y - r).abs() > 0.05 * r.abs().max()
            n = int(bad.sum())
            if n:
                runs_bad += 1
                nbad += n
                if where is None:
                    idx = bad.nonzero()
                    where = [idx[:, d].unique()[:8].tolist() for d in range(4)]
        print(f"mode {mode} ({'D=3' if mode == 1 else 'D=1'}) {name}: {runs_bad} of {REP} runs wrong, {nbad} elements; first at {where}", flush=True)
    for it in range(REP // 4 if len(fs) == 1 else 0):
        y3, mean, var = ops.resize_conv3x3_fwd_sum_bn(zs, (H, W), addvec=add)
        yf = y3.float().reshape(-1, N)
        bad = (y3.float() - (refp + add)).abs() > 0.05 * ref.abs().max()
        em = (mean - yf.mean(0)).abs().max().item()
        ev = (var - yf.var(0, unbiased=False)).abs().max().item()
        if int(bad.sum()) or em > 1e-3 or ev > 1e-2:
            print(f"mode {mode} statistics run {it}: bad outputs {int(bad.sum())}, mean err {em:.2e}, var err {ev:.2e}", flush=True)
lib.gdl_debug_set_tapsum_roll(1)
