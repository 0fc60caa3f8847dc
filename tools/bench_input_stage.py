#!/usr/bin/env python
"""Where does the PCIe-inclusive training rate lose against the HBM-resident one?  Same DOFA train step (batch 32, bf16), fed
(a) by one resident batch, (b)-(e) by host uint8 tiles + int64 masks through DeviceInputStage in several configurations, and
(f) the stage alone (no model): its own tiles/s.   python tools/bench_input_stage.py [steps]"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "geo-deep-learning_amd")]
sys.argv, args = sys.argv[:1], sys.argv[1:]
import bench  # noqa: E402
from geo_deep_learning.datamodules.device_input import DeviceInputStage  # noqa: E402

steps = int(args[0]) if args else 60
B = 32
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
task, opt = bench.build_task("dofa", dev, False, 0)
resident = bench.synthetic_batch(B, dev, 42, "dofa")
g = torch.Generator(device="cpu").manual_seed(7)
host = [{"image": torch.randint(0, 256, (B, 3, 512, 512), generator=g, dtype=torch.uint8),
         "mask": torch.randint(0, 5, (B, 1, 512, 512), generator=g, dtype=torch.int64),
         "wavelengths": torch.tensor(bench.WAVELENGTHS),
         "mean": torch.tensor(bench.RGB_MEAN).view(1, 3, 1, 1).expand(B, 3, 1, 1).contiguous(),
         "std": torch.tensor(bench.RGB_STD).view(1, 3, 1, 1).expand(B, 3, 1, 1).contiguous()} for _ in range(3)]


def run(get_batch, n):
    train, _ = bench.make_steps(task, opt, get_batch, True)
    for _ in range(5):
        train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        train()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def staged(**kw):
    st = DeviceInputStage((host[i % 3] for i in range(steps + 5)), dev, **kw)
    it = iter(st)
    return run(lambda: next(it), steps)


print(f"resident batch            : {run(lambda: resident, steps):7.2f} ms / step", flush=True)
for name, kw in (("worker thread (4 host threads), depth 2", dict(depth=2)), ("consumer thread, depth 2", dict(depth=2, threaded=False)),
                 ("worker thread, 1 host thread", dict(depth=2, host_threads=1)), ("worker thread, 128 host threads", dict(depth=2, host_threads=128)),
                 ("worker thread, int64 masks as they are", dict(depth=2, narrow_mask=False)),
                 ("worker thread (4 host threads) again", dict(depth=2))):
    print(f"{name:42s}: {staged(**kw):7.2f} ms / step", flush=True)
print(f"resident batch (again)    : {run(lambda: resident, steps):7.2f} ms / step", flush=True)
# pinned host batches (what a DataLoader(pin_memory=True) hands over): no copy into the ring
pinned = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) and k != "wavelengths" else v) for k, v in b.items()} for b in host]
st = DeviceInputStage((pinned[i % 3] for i in range(steps + 5)), dev, depth=2)
it = iter(st)
print(f"{'worker thread, source already pinned':42s}: {run(lambda: next(it), steps):7.2f} ms / step", flush=True)
# the stage on its own
for name, kw in (("worker thread", dict()), ("consumer thread", dict(threaded=False))):
    st = DeviceInputStage((host[i % 3] for i in range(steps)), dev, depth=2, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for b in st:
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"stage alone, {name:16s}: {dt / n * 1e3:7.2f} ms / batch = {B * n / dt:8.0f} tiles/s", flush=True)
