#!/usr/bin/env python
"""Where do the small torch launches of one DOFA training step come from?  One eager step (batch 32) under torch.profiler with
Python stacks; every aten op that launches a fill / copy / cast / add kernel is charged to the innermost frame inside this
repository.  Output: calls per step by (aten op, file:line)."""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd"))
sys.path.insert(0, str(ROOT / "geo-deep-learning_amd" / "geo_deep_learning"))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
batch_size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
task, optimizer = bench.build_task("dofa", dev, False, 0)
batch = bench.synthetic_batch(batch_size, dev, 42, "dofa")
train_step, _ = bench.make_steps(task, optimizer, lambda: batch, True)
for _ in range(3):
    train_step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    train_step()
    torch.cuda.synchronize()

WATCH = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::clone", "aten::_to_copy", "aten::contiguous", "aten::add",
         "aten::add_", "aten::mul", "aten::mul_", "aten::cat", "aten::empty_like", "aten::div", "aten::sum", "aten::sub", "aten::neg",
         "aten::zeros_like", "aten::full", "aten::ones", "aten::bernoulli_", "aten::_foreach_add_", "aten::index_select", "aten::stack")
by_site = collections.Counter()
dev_us = collections.Counter()
repo = str(ROOT)
for ev in prof.events():
    if ev.name not in WATCH or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    # only ops that reached the GPU (a kernel or memcpy child)
    t = sum(k.duration for k in getattr(ev, "kernels", []))
    if not getattr(ev, "kernels", None):
        continue
    site = "?"
    frames = list(ev.stack or [])
    for fr in frames:
        if ("gdlhip/" in fr or "geo_deep_learning/" in fr) and "tools/debug" not in fr:
            site = fr[fr.find("geo-deep-learning_amd/") + len("geo-deep-learning_amd/"):][:110] if "geo-deep-learning_amd/" in fr else fr[-110:]
            break
    if site == "?":
        site = " <- ".join(f[-60:] for f in frames[:3]) or "(no stack)"
    by_site[(ev.name, site)] += 1
    dev_us[(ev.name, site)] += t
print(f"one eager DOFA training step, batch {batch_size}: torch ops with a device launch, by innermost repository frame")
tot = 0
for (name, site), n in sorted(by_site.items(), key=lambda kv: -dev_us[kv[0]]):
    print(f"{n:4d} x {name:22s} {dev_us[(name, site)]:8.1f} us   {site}")
    tot += dev_us[(name, site)]
print(f"total device time of these launches: {tot:.1f} us")
