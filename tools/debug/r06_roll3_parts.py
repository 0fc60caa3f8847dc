"""Tuning probe (round 6): the three-source rolling gather-sum with parts switched off (results are wrong on purpose)."""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
from gdlhip import _lib, ops  # noqa: E402
lib = _lib.load()
lib.gdl_debug_set_tapsum_roll.argtypes = [ctypes.c_int]
B, N = 64, 256
zs = [torch.randn(B, 144 // f, 144 // f, 9 * N, device="cuda").to(torch.bfloat16) for f in (2, 4, 8)]
def timed(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode, name in ((1, "as is"), (16, "no matrix-core work"), (32, "no row fetches"), (64, "no stores"), (48, "no fetches, no compute"), (80, "no compute, no stores"), (96, "no fetches, no stores"), (112, "barriers and tile only"), (1, "as is")):
    lib.gdl_debug_set_tapsum_roll(mode)
    print(f"{name:28s} {timed(lambda: ops.resize_conv3x3_fwd_sum(zs, (144, 144))):8.1f} us", flush=True)
lib.gdl_debug_set_tapsum_roll(1)
