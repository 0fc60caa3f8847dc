"""hipGraph capture of whole steps (static shapes): one graph replay instead of several hundred kernel launches issued
one by one through ctypes.

At the reference's own per-GPU batch of 4 (configs/dofa_config_RGB.yaml:85) a DOFA-base + UperNet training step is ~600
launches of 5-20 us each: the GPU finishes them faster than the host can issue them (batch 2 and batch 4 took the same
9.7 ms).  A captured step has no host work per kernel.  Everything the step does runs on torch's current stream through
the C-ABI, allocates through torch's caching allocator (graph-private pool under capture) and keeps per-step scalars on
the device (``FusedAdam(capturable=True)``: step count, bias corrections, learning rate), so ``torch.cuda.graph`` can
record forward + loss + backward + clipping + Adam as they are.

Rules (torch.cuda.graphs): static input buffers (new batches are COPIED into them), no host read-backs inside the step, a
few eager warm-up steps on a side stream first (lazy initialisations, cached weight casts, kernel attributes).  DropPath /
Dropout2d draws use torch's graph-safe Philox generator: each replay draws fresh masks.

Under DistributedDataParallel (round 5; the reference's own deployment shape is `devices: -1` with per-GPU batch 4,
configs/dofa_config_RGB.yaml:5,13,85) the WHOLE step is captured, collectives included: RCCL's all-reduce / broadcast kernels
are ordinary stream work and torch's NCCL process group records them (and the event edges between its communication stream and
the compute stream) into the graph -- DDP's bucket all-reduces keep overlapping with the rest of backward inside the replay,
the SyncBatchNorm statistics messages sit where they sat, and a replay is one launch per rank.  What torch asks for
(CUDA-graphs notes, "Usage with DistributedDataParallel"): the DDP wrapper constructed on a side stream, >= 11 eager DDP
iterations before the capture (the reducer rebuilds its buckets after the first one and samples its run-time statistics with
events during the first ten), async error handling of the process group off.  Only the `nccl` backend can be captured (gloo
moves tensors through the host); ranks agree on the outcome of the capture with one all-reduce, so either all replay or all run
eagerly (see ``ddp_warmup`` / ``capturable_process_group`` below and MiniTrainer._graph_step).
"""

from __future__ import annotations

import contextlib
import gc

from typing import Any

import torch
from torch import Tensor

from . import nn as gnn


@contextlib.contextmanager
def _no_gc_during_capture():
    """Collect garbage BEFORE a stream capture and keep the cyclic collector off during it.  torch.cuda.graph no longer collects
    on entry (torch >= 2.6: only with torch.compiler.config.force_cudagraph_gc), so a collection triggered by the capture's own
    allocations could finalise earlier graphs, events or pinned buffers in the middle of it -- hipGraphExecDestroy / hipEventDestroy
    / hipHostFree inside a capture abort the process (seen in round 6: the DDP capture test died with `Aborted` in whole-suite
    order, depending on how much garbage the tests before it had left)."""
    gc.collect()
    was_on = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_on:
            gc.enable()


class GraphCaptureFatal(RuntimeError):
    """A HIP error INSIDE a stream capture (a host read-back, a synchronising call in the step): the capture is invalidated and
    torch leaves the CUDA generator in capture mode and the allocator routed to the dead graph's pool -- the process cannot be
    trusted to train on.  Raised instead of falling back to eager steps; rerun with graph_step=False."""


DDP_WARMUP = 11      # eager DDP iterations torch wants before a whole-network capture (see the module docstring)


def find_ddp(module: torch.nn.Module):
    """The DistributedDataParallel wrapper inside a task (``task.model`` under MiniTrainer / Lightning), or None."""
    for m in module.modules():
        if isinstance(m, torch.nn.parallel.DistributedDataParallel):
            return m
    return None


def capturable_process_group(group=None) -> bool:
    """True when collectives on ``group`` are stream work a hipGraph can record: the nccl (= RCCL) backend."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return True
    try:
        if str(dist.get_backend(group)).lower() != "nccl":
            return False
    except (RuntimeError, ValueError):
        return False
    import os
    if os.environ.get("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0") not in ("0", ""):
        # torch's notes on DDP under CUDA graphs: the watchdog's error handling must be off (geo_deep_learning/train.py and
        # bench.py set it before init_process_group); with it on, say why the step stays eager instead of dying in the capture
        import logging
        logging.getLogger(__name__).warning("TORCH_NCCL_ASYNC_ERROR_HANDLING=%s: the DDP training step is not captured into a "
                                            "hipGraph (set it to 0 before init_process_group)",
                                            os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"])
        return False
    return True


def ddp_on_side_stream(module: torch.nn.Module, **ddp_kwargs) -> torch.nn.parallel.DistributedDataParallel:
    """DistributedDataParallel(module) constructed under a dedicated side stream, kept as ``ddp.gdl_stream``.  DDP creates (and
    stashes) every parameter's AccumulateGrad node at construction and autograd runs such a node on the stream it was created
    on: a whole-backward capture only works when wrapper construction, warm-up iterations and the capture itself use ONE
    non-default stream (measured, round 5: with the wrapper built on one side stream and the capture on torch.cuda.graph's own,
    the capture dies with hipErrorStreamCaptureUnsupported).  GraphedTrainStep picks ``gdl_stream`` up; a trainer should run its
    eager DDP steps under it too (gradient producer and AccumulateGrad then share a stream: no cross-stream waits)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ddp = torch.nn.parallel.DistributedDataParallel(module, **ddp_kwargs)
    torch.cuda.current_stream().wait_stream(side)
    ddp.gdl_stream = side
    return ddp


def _clone_static(batch: dict[str, Any]) -> dict[str, Any]:
    return {k: (v.clone() if isinstance(v, Tensor) and v.is_cuda else v) for k, v in batch.items()}


def _copy_into(static: dict[str, Any], batch: dict[str, Any]) -> None:
    for k, v in batch.items():
        s = static.get(k)
        if isinstance(s, Tensor) and s.is_cuda:
            if not isinstance(v, Tensor) or v.shape != s.shape or v.dtype != s.dtype:
                msg = f"graphed step: batch[{k!r}] must keep shape {tuple(s.shape)} / dtype {s.dtype} of the captured batch"
                raise ValueError(msg)
            s.copy_(v, non_blocking=True)


class GraphedTrainStep:
    """``loss = step(batch)`` == zero_grad -> autocast(training_step) -> backward -> optimizer.step(), replayed from a hipGraph.

    ``task``: a LightningModule-shaped task (``training_step(batch, idx) -> loss``); ``optimizer``: ``FusedAdam`` created with
    ``capturable=True``.  The returned loss is a static device tensor (read it after the next synchronisation point)."""

    def __init__(self, task, optimizer, example_batch: dict[str, Any], *, autocast_dtype: torch.dtype | None = torch.bfloat16,
                 warmup: int = 3, restore_state: bool = False) -> None:
        """``restore_state``: the warm-up steps are REAL optimizer steps on ``example_batch`` (the capture pass only records).
        A trainer that captures in the middle of ``fit`` must not train three extra times on its first batch: with
        ``restore_state`` parameters, buffers, optimizer state (moments, step counts, device-side hyper-parameters), the bf16
        GEMM operands the optimizer kernel keeps in step with the parameters, and the CUDA RNG state are put back IN PLACE after
        the capture (the graph holds their addresses), so the first replay is the first training step."""
        if not getattr(optimizer, "capturable", False):
            msg = "GraphedTrainStep needs FusedAdam(capturable=True): step count and learning rate must live on the device"
            raise ValueError(msg)
        self.ddp = find_ddp(task)
        self.capture_stream = None
        if self.ddp is not None:
            if not capturable_process_group(self.ddp.process_group):
                msg = "GraphedTrainStep under DistributedDataParallel needs the nccl (RCCL) backend: gloo collectives pass through the host"
                raise ValueError(msg)
            self.capture_stream = getattr(self.ddp, "gdl_stream", None)
            if self.capture_stream is None:
                msg = ("GraphedTrainStep under DistributedDataParallel needs the wrapper built by gdlhip.graphs.ddp_on_side_stream "
                       "(wrapper construction, warm-up and capture must share one non-default stream)")
                raise ValueError(msg)
            warmup = max(warmup, DDP_WARMUP)
        self.task, self.optimizer, self.autocast_dtype = task, optimizer, autocast_dtype
        self.static = _clone_static(example_batch)
        task.train()
        # gradients (and their AccumulateGrad nodes) of earlier eager steps belong to another stream: start clean
        optimizer.zero_grad(set_to_none=True)
        if isinstance(getattr(task, "logged", None), dict):
            task.logged.clear()
        # a trainer's metric sink accumulates logged tensors on the device: captured once, it would add the SAME static buffer at
        # every replay while the host-side weights stood still.  During warm-up and capture the sink only RECORDS what the step
        # logs (name, static tensor, batch size); after every replay __call__ hands those tensors to the real sink
        trainer = getattr(task, "trainer", None)
        sink = getattr(trainer, "_collect", None)
        if sink is not None:
            trainer._collect = lambda *a, **k: None
        self.logged_static: list = []      # (name, static tensor, batch size) of every `log` call of the captured step
        if sink is not None:
            trainer._collect = lambda name, value, batch_size=None: self.logged_static.append((name, value, batch_size))
        # the snapshot is taken whenever the caller wants a clean fallback: a capture that raises after its warm-up steps has
        # already trained on `example_batch` (parameters, moments, step counts, BN running statistics, RNG all advanced) and its
        # gradients may live in a graph pool that dies with the failed graph -- both are undone before the exception leaves
        snap = self._snapshot(task, optimizer) if restore_state else None
        # host-side bookkeeping of training_step (sample counters) runs during the warm-up and the capture pass only: after the
        # capture `training_step` never executes on the host again.  The counters are put back here and advanced per replay by
        # the task's `on_graph_replay(batch)` hook (see __call__); training_step must otherwise be free of host side effects
        host_counters = {k: getattr(task, k) for k in ("train_samples_count",) if isinstance(getattr(task, k, None), int)}
        try:
            self._capture(task, optimizer, warmup)
        except BaseException as exc:
            # A Python exception inside the captured step: torch.cuda.graph's __exit__ has ended the capture (measured on gfx950:
            # the stream is out of capture, the partial graph is dropped) and the restoring copies below are ordinary eager work.
            # A HIP error inside the captured step (a host read-back, a synchronising call) is different: the capture is
            # INVALIDATED, capture_end raises before torch leaves its stream context, and HIP keeps reporting the stream as
            # capturing -- _leave_broken_capture puts the current stream back and ends the capture by hand; if the device still
            # refuses to synchronise the process cannot train on and says so instead of failing somewhere else later
            invalidated = self._leave_broken_capture()
            try:
                torch.cuda.synchronize()
            except RuntimeError as sync_exc:
                msg = ("hipGraph capture of the training step failed with a HIP error inside the capture and left the device in "
                       f"capture state ({type(sync_exc).__name__}); restart with graph_step=False.  Original error: {exc}")
                raise GraphCaptureFatal(msg) from exc
            optimizer.zero_grad(set_to_none=True)
            self.graph = None
            if snap is not None:
                self._restore(snap, optimizer)
            if invalidated:
                # the training state is back, but torch's capture_end never reached its epilogue: the CUDA generator still thinks a
                # capture is running (the next DropPath draw raises "Offset increment outside graph capture") -- measured on ROCm 7
                msg = ("hipGraph capture of the training step hit a HIP error inside the capture (a host read-back or a "
                       "synchronising call in training_step?).  Parameters, optimizer state and buffers were restored, but torch's "
                       "RNG / allocator capture state cannot be: restart with graph_step=False (MiniTrainer) or fix the step.  "
                       f"Original error: {type(exc).__name__}: {exc}")
                raise GraphCaptureFatal(msg) from exc
            raise
        finally:
            if sink is not None:
                trainer._collect = sink
            if restore_state:
                for k, v in host_counters.items():
                    setattr(task, k, v)
        if snap is not None:
            self._restore(snap, optimizer)

    def _leave_broken_capture(self) -> bool:
        """After capture_end raised (capture invalidated by a HIP error): torch.cuda.graph.__exit__ did not restore the current
        stream, and on ROCm 7 the capture stream stays in `invalidated` state.  Best effort: leave the stream context and call
        hipStreamEndCapture once more on the capture stream (it returns the invalidation error and resets the state)."""
        ctx, self._capture_ctx = getattr(self, "_capture_ctx", None), None
        if ctx is None or not torch.cuda.is_current_stream_capturing():
            return False
        stream = torch.cuda.current_stream()
        try:
            ctx.stream_ctx.__exit__(None, None, None)
        except Exception:  # noqa: BLE001
            pass
        try:      # the allocator still routes this stream's allocations to the dead graph's pool
            torch._C._cuda_endAllocateToPool(stream.device_index, self.graph.pool())
        except Exception:  # noqa: BLE001  (private API; the pool then simply stays alive)
            pass
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            graph = ctypes.c_void_p()
            hip.hipStreamEndCapture(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(graph))
            if graph.value:
                hip.hipGraphDestroy(graph)
            hip.hipGetLastError()
        except OSError:
            pass
        return True

    @staticmethod
    def _snapshot(task, optimizer):
        # what the warm-up steps can change: trainable parameters and buffers (BatchNorm running estimates, counters).  Frozen
        # parameters are left alone ON PURPOSE: restoring them would call mark_updated on them, which invalidates the operands
        # cached from them (the DOFA dynamic patch-embedding weights of a frozen encoder, bf16 weight copies) -- operands that were
        # VALID during the capture, i.e. whose addresses the graph holds.  The next eager step would rebuild and free them, and a
        # replay would read freed memory (round 5: found under DDP, where warm-up, capture and eager steps share one stream and the
        # allocator reuses the block at once; with the capture on torch's own side stream the block merely was never reused)
        tensors = [p for p in task.parameters() if p.requires_grad] + list(task.buffers())
        state = {}
        for p, st in optimizer.state.items():
            state[p] = {k: (v.detach().clone() if isinstance(v, Tensor) else v) for k, v in st.items()}
        dev = {gi: t.detach().clone() for gi, t in getattr(optimizer, "_dev_state", {}).items()}
        return {"tensors": [(t, t.detach().clone()) for t in tensors], "opt": state, "dev": dev,
                "rng": torch.cuda.get_rng_state(), "cpu_rng": torch.get_rng_state()}

    @staticmethod
    @torch.no_grad()
    def _restore(snap, optimizer) -> None:
        for t, saved in snap["tensors"]:
            t.copy_(saved)
            gnn.mark_updated(t)
        for p, st in optimizer.state.items():
            old = snap["opt"].get(p)
            for k, v in st.items():
                if isinstance(v, Tensor):
                    v.copy_(old[k]) if old is not None else v.zero_()
                elif k == "step":
                    st[k] = old[k] if old is not None else 0
        for gi, t in getattr(optimizer, "_dev_state", {}).items():
            if gi in snap["dev"]:
                t.copy_(snap["dev"][gi])
            else:
                # created during the warm-up: back to the step count the restored HOST state reports (what device_state() would
                # have seeded it with; 0 for a fresh optimizer, the loaded count after load_state_dict / an eager prefix) -- a
                # constant 0 would restart the bias correction at step 1 under warm moments
                steps = {st["step"] for p in optimizer.param_groups[gi]["params"]
                         if (st := optimizer.state.get(p)) and isinstance(st.get("step"), (int, float))}
                t[0] = float(steps.pop()) if len(steps) == 1 else 0.0
                t[6:8] = 0.0
        # the bf16 GEMM operands the optimizer kernel rewrites together with the parameters were captured by address and hold
        # the warm-up's values: rewrite them from the restored parameters (same element order by construction)
        for key, val, p in getattr(optimizer, "_shadowed", []):
            w = p.detach() if p.dim() == 2 else gnn.conv_weight_matrix(p)
            val.copy_(w.reshape(val.shape))
            gnn.refresh_shadow(key, val, p)
        # ... and so do the operands derived from them in another element order (channel slices, tap-major forms, data-gradient
        # operands): the capture must find them valid, or it would record their rebuild in front of every use
        if hasattr(optimizer, "refresh_derived"):
            optimizer.refresh_derived()
        torch.cuda.set_rng_state(snap["rng"])
        torch.set_rng_state(snap["cpu_rng"])

    def _capture(self, task, optimizer, warmup: int) -> None:
        side = self.capture_stream or torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.logged_static.clear()                      # (the warm-up steps logged eager tensors: only the captured ones count)
        optimizer.zero_grad(set_to_none=True)           # gradients are (re)allocated from the graph's private pool
        # under DDP the process group's watchdog thread polls the events of earlier collectives while this thread records:
        # "thread_local" restricts the capture-unsafe-call check to the capturing thread (backward still runs on the autograd
        # engine's device thread: work launched into a capturing stream is recorded whichever thread launches it).  The capture
        # runs on the stream the DDP wrapper was built on (see ddp_on_side_stream)
        mode = "thread_local" if self.ddp is not None else "global"
        ctx = torch.cuda.graph(self.graph, capture_error_mode=mode, **({"stream": self.capture_stream} if self.capture_stream else {}))
        self._capture_ctx = ctx
        with _no_gc_during_capture(), ctx:
            self.loss = self._eager(zero=False)
        self._capture_ctx = None
        # every operand cache entry that exists now may have been read by the captured kernels through its address (entries that
        # were valid during the capture were not rebuilt inside it): keep them alive as long as the graph
        self._cache_refs = [entry[1] for entry in list(gnn._CACHE.values())]
        # the bf16 operands the CAPTURED update rewrites in place (plain copies of the parameters and the operands derived from
        # them in another element order): current after every replay -- __call__ re-adopts them as the cache entries, so that an
        # eager step between two replays (a ragged last batch) reads and rewrites the SAME tensors instead of building its own
        # (found in round 6: the first replay after an eager bf16 step ran its forward on operands that missed that step's update)
        self._graph_operands = list(getattr(optimizer, "_shadowed", [])) + list(getattr(optimizer, "_derived_last", []))
        self._rewritten = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self._rewritten += [b for m in task.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training
                            for b in (m.running_mean, m.running_var) if b is not None]

    def _eager(self, zero: bool = True) -> Tensor:
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            loss = self.task.training_step(self.static, 0)
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self, batch: dict[str, Any] | None = None) -> Tensor:
        """Returns the STATIC loss buffer: the next replay overwrites it (clone it to keep a value)."""
        if batch is not None and batch is not self.static:
            _copy_into(self.static, batch)
        self.optimizer.sync_lr()        # per-step schedulers (OneCycleLR) write param_groups["lr"] on the host
        self.graph.replay()
        self.optimizer.note_replay()
        sink = getattr(getattr(self.task, "trainer", None), "_collect", None)
        if sink is not None:            # what the captured training_step logged: the static tensors now hold this step's values
            for name, value, batch_size in self.logged_static:
                sink(name, value, batch_size)
        for t in self._rewritten:       # the replay rewrote these through raw pointers: eager code must not trust operands
            gnn.mark_updated(t)         # cached from their earlier values (eval-time BatchNorm folds, packed weights)
        for key, val, p in self._graph_operands:     # ... except the ones the replay itself rewrote from the new parameters
            gnn.adopt_operand(key, val, p)
        hook = getattr(self.task, "on_graph_replay", None)
        if hook is not None:            # host bookkeeping the captured training_step can no longer do (sample counters)
            hook(self.static)
        return self.loss


class GraphedEvalStep:
    """``out = step(batch)`` == no_grad + autocast(fn(batch)) replayed from a hipGraph; ``fn`` e.g. ``lambda b: task(b["image"],
    b["wavelengths"])`` or a validation step that returns tensors (their storage is static: copy what must outlive a replay).
    The graph holds the ADDRESSES of the operands that were current at capture time (bf16 weight copies, BatchNorm-folded
    weights): it is for serving a model whose parameters no longer change -- capture again after training on."""

    def __init__(self, fn, example_batch: dict[str, Any], *, autocast_dtype: torch.dtype | None = torch.bfloat16, warmup: int = 2) -> None:
        self.fn, self.autocast_dtype = fn, autocast_dtype
        self.static = _clone_static(example_batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with _no_gc_during_capture(), torch.cuda.graph(self.graph):
            self.out = self._eager()

    def _eager(self):
        with torch.no_grad(), torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            return self.fn(self.static)

    def __call__(self, batch: dict[str, Any] | None = None):
        if batch is not None and batch is not self.static:
            _copy_into(self.static, batch)
        self.graph.replay()
        return self.out
