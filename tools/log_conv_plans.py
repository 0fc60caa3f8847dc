#!/usr/bin/env python
"""Which implicit-GEMM tile does every convolution / linear of one training step get, and how long does it take?
Wraps gdlhip.ops.conv_gemm for ONE train step of a bench.py model (after two warm-up steps) and prints, per distinct
call shape, the planner's variant (0 = 64^2, 1 = 128^2, 3 = 256^2 ping-pong, 4 = 3x3 shared staging, 5 = 256 x 64),
the call count, the HIP-event time and the TF/s.   tools/log_conv_plans.py [dofa|segformer|unetpp] [batch]"""
import collections
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
import bench  # noqa: E402
from gdlhip import ops  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "dofa"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
task, optimizer = bench.build_task(model, dev, False, 0)
batch = bench.synthetic_batch(B, dev, 43)
train_step, _ = bench.make_steps(task, optimizer, lambda: batch, True)
for _ in range(2):
    train_step()
torch.cuda.synchronize()

orig = ops.conv_gemm
rows = collections.OrderedDict()


def logged(x, w, **kw):
    R, S, stride, pad = kw.get("R", 1), kw.get("S", 1), kw.get("stride", 1), kw.get("pad", 0)
    x4 = x if x.dim() == 4 else x.reshape(1, 1, -1, x.shape[-1])
    Bq, H, W, Cc = x4.shape
    N = w.shape[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ret = orig(x, w, **kw)
    e1.record()
    out = ret[0] if isinstance(ret, tuple) else ret          # want_stats=True returns (out, partials, rows)
    o4 = out if out.dim() == 4 else out.reshape(1, 1, -1, out.shape[-1])
    key = (Bq, H, W, Cc, N, R, S, stride, pad, str(x.dtype).replace("torch.", ""), str(out.dtype).replace("torch.", ""),
           "resid" if kw.get("resid") is not None else "", o4.shape[1] * o4.shape[2] * Bq)
    rows.setdefault(key, []).append((e0, e1))
    return ret


orig_w = ops.conv_wgrad
wrows = collections.OrderedDict()


def logged_w(x, dy, **kw):
    x4 = x if x.dim() == 4 else x.reshape(1, 1, -1, x.shape[-1])
    d4 = dy if dy.dim() == 4 else dy.reshape(1, 1, -1, dy.shape[-1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_w(x, dy, **kw)
    e1.record()
    key = (*x4.shape, d4.shape[-1], kw.get("R", 1), kw.get("S", 1), kw.get("stride", 1), d4.shape[0] * d4.shape[1] * d4.shape[2])
    wrows.setdefault(key, []).append((e0, e1))
    return out


ops.conv_gemm = logged
ops.conv_wgrad = logged_w
train_step()
torch.cuda.synchronize()
ops.conv_gemm = orig
ops.conv_wgrad = orig_w
lib = ops._lib.load()
print(f"{model}, batch {B}: conv_gemm calls of one training step")
print(f"{'calls':>5} {'B':>3} {'H':>4} {'W':>6} {'C':>5} {'N':>5} RxS/s/p   {'M':>8} {'us/call':>8} {'TF/s':>7}  in->out")
tot = 0.0
for k, evs in rows.items():
    Bq, H, W, Cc, N, R, S, stride, pad, idt, odt, res, M = k
    us = sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3
    tot += us * len(evs)
    fl = 2.0 * M * N * R * S * Cc
    print(f"{len(evs):5d} {Bq:3d} {H:4d} {W:6d} {Cc:5d} {N:5d} {R}x{S}/{stride}/{pad} {M:10d} {us:8.1f} {fl / us / 1e6:7.1f}  {idt}->{odt} {res}")
print(f"total {tot / 1e3:.2f} ms in conv_gemm (HIP events around single calls: includes launch gaps)")
print(f"{model}, batch {B}: conv_wgrad calls of one training step (pixels are the reduction dimension)")
print(f"{'calls':>5} {'B':>3} {'H':>4} {'W':>6} {'C':>5} {'N':>5} RxS/s  {'pixels':>8} {'us/call':>8} {'TF/s':>7}")
tot = 0.0
for k, evs in wrows.items():
    Bq, H, W, Cc, N, R, S, stride, M = k
    us = sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3
    tot += us * len(evs)
    print(f"{len(evs):5d} {Bq:3d} {H:4d} {W:6d} {Cc:5d} {N:5d} {R}x{S}/{stride} {M:10d} {us:8.1f} {2.0 * M * N * R * S * Cc / us / 1e6:7.1f}")
print(f"total {tot / 1e3:.2f} ms in conv_wgrad (incl. split-K reductions)")
