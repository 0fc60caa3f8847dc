"""GPU-side input stage.

The reference's loader (datamodules/wds_datamodule.py:104-111: ``WebLoader(pin_memory=True)``) hands Lightning
f32 batches that were normalised by CPU workers; Lightning then copies them synchronously.  This stage sits
between any iterable of host batch dicts and the training loop:

* every host tensor is staged through a ring of **pinned** buffers and copied with ``non_blocking=True`` on a
  dedicated HIP copy stream, ``depth`` batches ahead of the consumer -- the copy of batch k+1 overlaps the
  compute of batch k; the host side of the staging runs on a worker thread (round 5), so it also overlaps the
  consumer thread's kernel launches;
* raw tiles (``image`` in uint8 / uint16 / int16 / f32 as stored) are normalised **on the GPU** by one
  HBM-bound kernel (``x/255 -> (x-mean)/std``, gdl_normalize_raw) on the compute stream, after it waited for the
  copy event: uint8 tiles cross PCIe at 1 byte / sample instead of the reference's 4;
* with ``augment=gdlhip.augment.reference_pipeline(size)`` the reference's kornia augmentations (main-process CPU,
  segmentation_dofa.py:201-211) run in that same kernel;
* int64 masks whose values fit a byte (class indices) cross the link as uint8 and are widened again on the device:
  2.1 of the 2.9 MB a 512 x 512 RGB tile + mask used to ship were the mask's upper seven bytes;
* the yielded dict has the reference's keys and dtypes (``image`` f32 standardised, ``mask`` int64, ...), so
  ``training_step`` / ``validation_step`` are unchanged.
"""

from __future__ import annotations

from collections import deque
from collections.abc import Iterable, Iterator
from typing import Any

import torch

from gdlhip import ops


class DeviceInputStage:
    """Iterate device-resident, normalised batches ``depth`` copies ahead of the consumer."""

    def __init__(self, batches: Iterable[dict[str, Any]], device: torch.device | str = "cuda", depth: int = 2,
                 raw_key: str = "image", augment: Any | None = None, narrow_mask: bool = True, threaded: bool = True,
                 host_threads: int = 4) -> None:
        self.batches = batches
        self.device = torch.device(device)
        if self.device.type != "cuda":
            msg = "DeviceInputStage needs a HIP device (no CPU fallback)"
            raise ValueError(msg)
        self.depth = max(1, int(depth))
        self.raw_key = raw_key
        self.augment = augment      # gdlhip.augment.AugmentationSequential: fused with the normalise kernel
        self.narrow_mask = narrow_mask   # int64 masks with values in 0..255 are copied as uint8
        self.threaded = threaded         # stage on a worker thread (False: on the consumer thread, as in round 4)
        # OpenMP threads of the worker's copy / range-check kernels.  torch's default is one per hardware thread (128 on the GPU
        # boxes), all of which spin for a while after every parallel region; in a CPU-quota'd container that spinning throttles
        # the thread that launches the training step (round 5: the staged step took 67-93 ms against 35.7 ms resident)
        self.host_threads = max(1, int(host_threads))
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned: dict = {}      # (key, shape, dtype) -> ring of pinned host buffers
        self._slot_events: list = [None] * (self.depth + 1)   # copy-done event of the batch that last used a ring slot
        self._slot = 0
        import threading
        self._stage_lock = threading.Lock()                   # serialises _stage between a finishing and a starting worker thread
        self.bytes_h2d = 0

    # ------------------------------------------------------------------ staging
    def _pinned_buffer(self, key: str, t: torch.Tensor) -> torch.Tensor:
        sig = (key, tuple(t.shape), t.dtype)
        ring = self._pinned.get(sig)
        if ring is None:
            ring = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for _ in range(self.depth + 1)]
            self._pinned[sig] = ring
        return ring[self._slot % (self.depth + 1)]

    def _stage(self, batch: dict[str, Any]):
        """Enqueue the H2D copies of one batch on the copy stream; returns (device dict, event)."""
        dev: dict[str, Any] = {}
        # the pinned buffers of this ring slot were the SOURCE of an asynchronous H2D copy depth + 1 batches ago: the host
        # must not overwrite them before that copy has finished (only the compute stream waits on the event otherwise)
        slot = self._slot % (self.depth + 1)
        if self._slot_events[slot] is not None:
            self._slot_events[slot].synchronize()
        with torch.cuda.stream(self._copy_stream):
            for k, v in batch.items():
                if isinstance(v, torch.Tensor) and not v.is_cuda and k != "wavelengths":
                    if k == "mask" and self.narrow_mask and v.dtype == torch.int64 and v.numel() and self._fits_a_byte(v):
                        src = self._pinned_buffer(k + ":u8", torch.empty(0, dtype=torch.uint8).new_empty(v.shape))
                        src.copy_(v)                                   # narrowing cast on the host, into the pinned ring
                        dev[k] = src.to(self.device, non_blocking=True)
                        dev["_mask_narrowed"] = True
                        self.bytes_h2d += v.numel()
                        continue
                    if v.is_pinned():
                        src = v
                    else:
                        src = self._pinned_buffer(k, v)
                        src.copy_(v)
                    dev[k] = src.to(self.device, non_blocking=True)
                    self.bytes_h2d += v.numel() * v.element_size()
                else:
                    dev[k] = v        # wavelengths stay on the host (the encoder reads them there), strings, ...
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._slot_events[slot] = ev
        self._slot += 1
        return dev, ev

    @staticmethod
    def _fits_a_byte(v: torch.Tensor) -> bool:
        """All values in 0..255?  ONE pass (torch.aminmax: 0.8 ms for the 67 MB of 32 int64 masks); round 4 called amin() and
        amax(), whose int64 CPU kernels are not vectorised -- 99 ms for the same tensor on 8 threads, and the whole 10 % gap
        between the PCIe-inclusive and the HBM-resident training rate."""
        lo, hi = torch.aminmax(v)
        return int(lo) >= 0 and int(hi) <= 255

    def _finish(self, dev: dict[str, Any], ev: torch.cuda.Event) -> dict[str, Any]:
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for v in dev.values():
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(cur)  # allocated on the copy stream, consumed on the compute stream
        if dev.pop("_mask_narrowed", False):
            dev["mask"] = dev["mask"].to(torch.int64)      # the reference's dtype again (compute stream, after the copy event)
        img = dev.get(self.raw_key)
        raw = isinstance(img, torch.Tensor) and self._is_raw(img, dev)
        if self.augment is not None and isinstance(img, torch.Tensor):
            mean, std = self._sensor_stats(dev) if raw else (None, None)
            dev = self.augment(dev, mean, std)          # ONE kernel: normalise each tap + flip / rot90 / crop-resize
        elif raw:
            mean, std = self._sensor_stats(dev)
            dev[self.raw_key] = ops.normalize_raw(img.contiguous(), mean, std)
        return dev

    @staticmethod
    def _is_raw(img: torch.Tensor, dev: dict[str, Any]) -> bool:
        if img.dtype in (torch.uint8, torch.uint16, torch.int16):
            return True
        return bool(dev.get("image_is_raw", False))

    @staticmethod
    def _sensor_stats(dev: dict[str, Any]) -> tuple[torch.Tensor, torch.Tensor]:
        """[C] f32 device vectors from the batch's ``mean`` / ``std`` ([C,1,1] or stacked [B,C,1,1]; one sensor
        per batch, as the reference batches per sensor dataset)."""
        out = []
        for k in ("mean", "std"):
            t = dev[k]
            c = dev["image"].shape[1]
            t = t.reshape(-1, c)[0] if t.numel() != c else t.reshape(c)
            out.append(t.to(dtype=torch.float32).contiguous())
        return out[0], out[1]

    # ------------------------------------------------------------------ iteration
    def __iter__(self) -> Iterator[dict[str, Any]]:
        if self.threaded:
            yield from self._iter_threaded()
            return
        queue: deque = deque()
        it = iter(self.batches)
        for _ in range(self.depth):
            try:
                queue.append(self._stage(next(it)))
            except StopIteration:
                break
        while queue:
            dev, ev = queue.popleft()
            try:
                queue.append(self._stage(next(it)))     # keep the copy stream `depth` batches ahead
            except StopIteration:
                pass
            yield self._finish(dev, ev)

    def _iter_threaded(self) -> Iterator[dict[str, Any]]:
        """The staging (range check, narrowing cast and copies into the pinned ring: 2-15 ms of host time per 32-tile batch,
        depending on how busy the box's cores are) runs on a worker thread, ``depth`` batches ahead; the consumer thread -- the one
        that issues the training step's ~470 launches -- only waits on the copy event.  torch's CPU kernels release the GIL, so
        the two overlap; round 4 staged on the consumer thread and lost 10-35 % of the HBM-resident rate whenever staging +
        launching took longer than the GPU step."""
        import queue as queue_mod
        import threading
        q: queue_mod.Queue = queue_mod.Queue(maxsize=self.depth)
        # (a new thread starts on device 0: it gets the consumer's device -- "cuda" without an index means the current one HERE)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        stop = threading.Event()
        done = object()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue_mod.Full:
                    continue
            return False

        def work() -> None:
            # torch.set_num_threads stores a PROCESS-wide default that threads created later inherit (the autograd engine's,
            # later workers'), not just this thread's OpenMP team: put the previous value back on the way out (advisor, round 5)
            prev_threads = torch.get_num_threads()
            try:
                torch.cuda.set_device(dev_index)
                torch.set_num_threads(self.host_threads)
                for batch in self.batches:
                    if stop.is_set():
                        return
                    with self._stage_lock:                    # one stager at a time touches the pinned ring and its slot counter
                        staged = self._stage(batch)
                    if not put(staged):
                        return
                put(done)
            except BaseException as exc:  # noqa: BLE001  (handed to the consumer)
                put(exc)
            finally:
                torch.set_num_threads(prev_threads)

        worker = threading.Thread(target=work, name="gdl-input-stage", daemon=True)
        worker.start()
        try:
            while True:
                item = q.get()
                if item is done:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self._finish(*item)
        finally:
            # an early exit of the consumer (limit_*_batches, fast_dev_run, an exception in the step): stop the worker, empty
            # the queue so that a blocked put() returns, and WAIT for it -- a worker still inside `for batch in self.batches` /
            # _stage when the stage is iterated again would share the loader, the slot counter and the pinned ring with the
            # next worker and could overwrite a pinned slot whose copy is still in flight (advisor, round 5)
            stop.set()
            while worker.is_alive():
                try:
                    q.get_nowait()
                except queue_mod.Empty:
                    pass
                worker.join(timeout=0.05)

    def __len__(self) -> int:
        return len(self.batches)  # type: ignore[arg-type]
