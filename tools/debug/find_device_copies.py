#!/usr/bin/env python
"""Where do the small device-to-device copies of a step come from?  One inference (and one training) step of the benchmarked
DOFA model under torch.profiler with Python stacks; prints every Memcpy DtoD / copy kernel with the innermost repo frames.

    python tools/debug/find_device_copies.py [batch]"""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
task, opt = bench.build_task("dofa", dev, False, 0)
batch = bench.synthetic_batch(B, dev, 43, "dofa")
train_step, infer_step = bench.make_steps(task, opt, lambda: batch, True)
for name, fn in (("inference", infer_step), ("train", train_step)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
        fn()
        torch.cuda.synchronize()
    sites = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::cat", "aten::fill_", "aten::zero_", "aten::zeros", "aten::add", "aten::add_", "aten::mul", "aten::flip"):
            frames = [f for f in (ev.stack or []) if "geo-deep-learning_amd" in f or "bench.py" in f]
            sites[(ev.name, tuple(f.split("geo-deep-learning_amd/")[-1][:90] for f in frames[:2]))] += 1
    mem = collections.Counter()
    for ev in prof.events():
        if "emcpy" in ev.name or "emset" in ev.name or "copyBuffer" in ev.name or "fillBuffer" in ev.name:
            mem[ev.name[:60]] += 1
    print(f"== {name} step: runtime copies / fills:", dict(mem))
    cpu_ops = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::") and ev.name in (
        "aten::item", "aten::_local_scalar_dense", "aten::to", "aten::tensor", "aten::lift_fresh", "aten::empty", "aten::as_strided", "aten::copy_", "aten::_to_copy"))
    print("   host-visible aten ops:", dict(cpu_ops))
    print(f"== {name} step, batch {B}: aten ops that launch small kernels, by call site")
    for (op, fr), n in sites.most_common(40):
        print(f"  {n:4d} x {op:18s} {' <- '.join(fr)}")
