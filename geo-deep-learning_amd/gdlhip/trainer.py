"""MiniTrainer: the part of ``lightning.pytorch.Trainer`` that the reference's ``train.py fit`` exercises
(/root/reference/geo_deep_learning/train.py:27-80, configs/dofa_config_RGB.yaml:1-33,62-77), for images where Lightning is
not installed.  It drives the task classes through exactly the hooks Lightning would call:

    configure_model -> configure_optimizers -> per batch: on_before_batch_transfer -> (H2D) -> on_after_batch_transfer ->
    training_step -> backward -> clip -> optimizer step; per epoch: validation_step, scheduler (ReduceLROnPlateau on the
    monitored metric), best-``val_loss`` checkpoint with the state dict under ``model.*`` keys; after fit: ``test`` of the
    best checkpoint on rank 0 (``GeoDeepLearningCLI.after_fit``).

One process per GPU: under ``torchrun`` the inner ``model.model`` is wrapped in ``DistributedDataParallel`` (RCCL =
backend "nccl"; "gloo" for the CPU tests), ``sync_batchnorm`` converts the BatchNorm containers, metrics are averaged
over ranks.  ``torch.optim.Adam`` from ``configure_optimizers`` is executed by the fused multi-tensor Adam + clip kernels
(``gdlhip.nn.FusedAdam``) on the SAME param groups, so schedulers keep working; any other optimizer runs as is.
Host logic only -- no tensor arithmetic of the hot path lives here.
"""

from __future__ import annotations

import contextlib
import logging
import math
import os
import random
import traceback
from pathlib import Path
from typing import Any

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor, nn

from .markers import rng

logger = logging.getLogger(__name__)

try:  # pragma: no cover - lightning is absent in the build image
    from lightning.pytorch import LightningModule
except ImportError:
    class LightningModule(nn.Module):  # type: ignore[no-redef]
        """Minimal stand-in with the surface the task hooks use (``log``, ``log_dict``, ``hparams``, ``trainer``,
        ``device``, ``save_hyperparameters``)."""

        def __init__(self) -> None:
            super().__init__()
            self.hparams: dict[str, Any] = {}
            self.trainer = None
            self.logged: dict[str, Any] = {}

        def save_hyperparameters(self, **kw: Any) -> None:
            self.hparams.update(kw)

        def log(self, name: str, value: Any, batch_size: int | None = None, **_kw: Any) -> None:
            # detached: a logged loss must not keep its autograd graph (and the AccumulateGrad nodes of every parameter) alive
            self.logged[name] = value.detach() if isinstance(value, Tensor) else value
            sink = getattr(self.trainer, "_collect", None)
            if sink is not None:
                sink(name, value, batch_size)

        def log_dict(self, d: dict[str, Any], batch_size: int | None = None, **_kw: Any) -> None:
            for k, v in d.items():
                self.log(k, v, batch_size=batch_size)

        @property
        def device(self) -> torch.device:
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")


def seed_everything(seed: int = 42, workers: bool = True) -> int:  # noqa: ARG001, FBT001, FBT002
    """lightning.pytorch.seed_everything (train.py:67)."""
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


class _Checkpoint:
    def __init__(self) -> None:
        self.best_model_path = ""
        self.best_model_score: float | None = None


# batch entries the models consume ON THE HOST (the DOFA encoder keys its dynamic patch-embedding weights by the wavelength
# values): moved to the GPU they would be read back with a blocking copy in every step -- and a read-back cannot be captured
HOST_BATCH_KEYS = ("wavelengths",)


def _to_device(obj: Any, device: torch.device) -> Any:
    if isinstance(obj, Tensor):
        return obj if obj.device == device else obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: (v if k in HOST_BATCH_KEYS and isinstance(v, Tensor) and not v.is_cuda else _to_device(v, device))
                for k, v in obj.items()}
    if isinstance(obj, (list, tuple)) and obj and isinstance(obj[0], Tensor):
        return type(obj)(_to_device(v, device) for v in obj)
    return obj


class MiniTrainer:
    """See the module docstring.  Constructor keywords follow ``lightning.pytorch.Trainer`` for the options the
    reference's configs set; unknown ones are accepted and ignored (logger / strategy objects, ...)."""

    def __init__(self, max_epochs: int = 10, precision: str | int = "32", gradient_clip_val: float | None = None,
                 sync_batchnorm: bool = False, accelerator: str = "auto", devices: Any = "auto",  # noqa: ARG002
                 default_root_dir: str | None = None, accumulate_grad_batches: int = 1,
                 limit_train_batches: int | None = None, limit_val_batches: int | None = None,
                 limit_test_batches: int | None = None, monitor: str = "val_loss", mode: str = "min",
                 checkpoint_filename: str = "model-{epoch:02d}-{val_loss:.3f}", early_stopping_patience: int | None = None,
                 use_fused_adam: bool = True, fast_dev_run: bool = False, graph_step: str | bool = "auto",
                 force_ddp: bool = False, **ignored: Any) -> None:
        self.max_epochs, self.precision = max_epochs, str(precision)
        self.gradient_clip_val, self.sync_batchnorm = gradient_clip_val, sync_batchnorm
        self.accumulate_grad_batches = max(1, int(accumulate_grad_batches))
        self.default_root_dir = Path(default_root_dir or os.getcwd())
        self.limit = {"train": limit_train_batches, "val": limit_val_batches, "test": limit_test_batches}
        if fast_dev_run:
            self.max_epochs, self.limit = 1, {"train": 1, "val": 1, "test": 1}
        self.monitor, self.mode = monitor, mode
        self.checkpoint_filename, self.early_stopping_patience = checkpoint_filename, early_stopping_patience
        self.use_fused_adam = use_fused_adam
        # hipGraph replay of the whole training step (gdlhip.graphs.GraphedTrainStep): "auto" = when the step is launch-bound
        # (per-GPU batch <= 8, e.g. the reference's own batch 4, configs/dofa_config_RGB.yaml:85: ~550 launches of 5-20 us) and
        # its shapes are static; True = whenever it can be captured; False = never.  Batches of another shape (a ragged last
        # batch) run eagerly through the same optimizer.
        # Under DDP (round 5) the captured step includes the collectives; only the nccl (RCCL) backend can be recorded, the ranks
        # agree on preconditions and outcome through the process group's store (_agree), and a failed capture leaves the training
        # state untouched (gdlhip.graphs).  With MORE THAN ONE rank "auto" keeps the step eager (round 6): that capture asks
        # for graph_step=True.
        self.graph_step = graph_step
        self.graph_max_batch = 8
        self.force_ddp = force_ddp           # wrap in DDP even with ONE rank (the single-GPU RCCL tests of the capture path)
        self.graphed_steps = 0               # how many training steps were graph replays (tests / logs)
        self._graphed = None
        if ignored:
            logger.info("MiniTrainer: ignoring trainer options %s", sorted(ignored))
        self.checkpoint_callback = _Checkpoint()
        self.training = False
        self.datamodule = None
        self.current_epoch = 0
        self.global_step = 0
        self.estimated_stepping_batches = -1
        self.callback_metrics: dict[str, float] = {}
        self.should_stop = False
        self._sums: dict[str, list[float]] = {}
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.global_rank = dist.get_rank() if self.world_size > 1 else 0

    # ------------------------------------------------------------------ small helpers
    @property
    def is_global_zero(self) -> bool:
        return self.global_rank == 0

    def _collect(self, name: str, value: Any, batch_size: int | None) -> None:
        """Sink of ``LightningModule.log``: batch-size weighted epoch means (on_epoch=True semantics)."""
        w = float(batch_size or 1)
        s = self._sums.setdefault(name, [0.0, 0.0])
        # tensors are accumulated on their device (no .item(): that would synchronise the stream on every step); the
        # host read-back happens once per epoch in _epoch_means
        v = value.detach().to(torch.float64) if isinstance(value, Tensor) else float(value)
        s[0] = s[0] + v * w
        s[1] += w

    def _epoch_means(self, device: torch.device) -> dict[str, float]:
        names = sorted(self._sums)
        if self.world_size > 1:
            # the reduction must have the same shape on every rank, whatever each rank logged: a rank whose slice of the
            # validation shards is empty logs nothing and would otherwise skip (or mis-size) the collective the others are in
            gathered: list = [None] * self.world_size
            dist.all_gather_object(gathered, names)
            names = sorted(set().union(*gathered))
        zero = [0.0, 0.0]
        vals = torch.stack([torch.stack([torch.as_tensor(self._sums.get(n, zero)[0], dtype=torch.float64, device=device).reshape(()),
                                         torch.tensor(self._sums.get(n, zero)[1], dtype=torch.float64, device=device)])
                            for n in names]) if names else torch.zeros((0, 2), dtype=torch.float64, device=device)
        if self.world_size > 1 and len(names):
            dist.all_reduce(vals)                        # sync_dist=True
        out = {n: (vals[i, 0] / vals[i, 1].clamp_min(1e-12)).item() for i, n in enumerate(names)}
        self._sums = {}
        return out

    def _autocast(self, device: torch.device):
        if self.precision in ("bf16-mixed", "bf16", "16-mixed", "16") and device.type == "cuda":
            # fp16-mixed of the reference's config maps to bf16 here (BASELINE.json: bf16; no loss scaler needed)
            if self.precision in ("16-mixed", "16") and not getattr(self, "_warned_fp16", False):
                logger.warning("precision=%s: the HIP kernels compute in bf16 (f32 accumulation); fp16 autocast + GradScaler "
                               "of the reference's config is replaced by bf16 autocast without loss scaling", self.precision)
                self._warned_fp16 = True
            return torch.autocast("cuda", dtype=torch.bfloat16)
        return torch.autocast(device.type, enabled=False)

    @staticmethod
    def _inner(model: nn.Module) -> nn.Module:
        m = getattr(model, "model", None)
        return m.module if isinstance(m, nn.parallel.DistributedDataParallel) else m

    def _device(self) -> torch.device:
        if torch.cuda.is_available():
            return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        return torch.device("cpu")

    def _loaders(self, datamodule, train=None, val=None):
        if datamodule is not None:
            if hasattr(datamodule, "prepare_data"):
                datamodule.prepare_data()
            datamodule.setup("fit")
            return datamodule.train_dataloader(), datamodule.val_dataloader()
        return train, val

    # ------------------------------------------------------------------ checkpoints
    def state_dict_of(self, model: nn.Module) -> dict[str, Tensor]:
        """Lightning layout: the LightningModule's state dict, i.e. the inner model's keys under ``model.`` (what
        utils/models.py:33 strips again).  The DDP wrapper's ``module.`` level is removed."""
        inner = self._inner(model)
        sd = {f"model.{k}": v.detach().cpu() for k, v in inner.state_dict().items()}
        for k, v in model.state_dict().items():
            if not k.startswith("model."):
                sd[k] = v.detach().cpu()
        return sd

    def save_checkpoint(self, model: nn.Module, path: str | Path, **extra: Any) -> None:
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        hp = {k: v for k, v in getattr(model, "hparams", {}).items() if isinstance(v, (int, float, str, bool, list, tuple, type(None), dict))}
        # the keys lightning's checkpoint migration / load_from_checkpoint look at (after_fit of the reference loads the best
        # checkpoint with the LightningModule's load_from_checkpoint): version tag, loops-free resume state
        ckpt = {"state_dict": self.state_dict_of(model), "epoch": self.current_epoch, "global_step": self.global_step,
                "pytorch-lightning_version": "2.5.0", "hyper_parameters": hp,
                "optimizer_states": [o.state_dict() for o in getattr(self, "_optimizers", [])],
                "lr_schedulers": [c["scheduler"].state_dict() for c in getattr(self, "_sched_cfgs", [])], **extra}
        torch.save(ckpt, path)

    def load_checkpoint(self, model: nn.Module, path: str | Path) -> None:
        ckpt = torch.load(path, map_location="cpu")
        sd = {k.removeprefix("model."): v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
        self._inner(model).load_state_dict(sd, strict=True)

    # ------------------------------------------------------------------ fit
    def fit(self, model: nn.Module, datamodule=None, train_dataloaders=None, val_dataloaders=None) -> None:
        device = self._device()
        model.trainer = self
        self.datamodule = datamodule
        model.configure_model()
        model.to(device)
        train_loader, val_loader = self._loaders(datamodule, train_dataloaders, val_dataloaders)
        want_graph = self.graph_step not in (False, "off", "false", None) and device.type == "cuda" and self.accumulate_grad_batches == 1
        self._stream = None
        self._ddp_active = self.world_size > 1 or (self.force_ddp and dist.is_available() and dist.is_initialized())
        if want_graph and self._ddp_active and self.world_size > 1 and self.graph_step == "auto":
            # The whole-step capture with RCCL collectives inside has run on ONE-rank groups and on two ranks sharing a GPU over
            # gloo only (no box with two devices was available): "auto" does not bet a multi-GPU training run on it.  A failed
            # capture there is not a fallback but a dead process group.  graph_step=True asks for it explicitly
            # (tests/test_hip_tasks.py::test_ddp_captured_step_world2_rccl_* run it wherever two devices exist).
            logger.info("MiniTrainer: graph_step='auto' keeps the DDP step eager at world size %d; pass graph_step=True to capture "
                        "it (RCCL collectives inside the hipGraph)", self.world_size)
            want_graph = False
        if self._ddp_active:
            if self.sync_batchnorm:
                model.model = nn.SyncBatchNorm.convert_sync_batchnorm(model.model)
            # device_ids stays None (legal for a module that lives on one device): with device_ids set, DDP's forward moves EVERY
            # input tensor to that device -- the batch's host-side `wavelengths` (HOST_BATCH_KEYS) too, which the DOFA encoder then
            # reads back with a blocking copy in every step, and which a hipGraph capture cannot record at all (measured, round 5)
            ddp_kw = dict(gradient_as_bucket_view=True, find_unused_parameters=False)
            if want_graph:
                from gdlhip.graphs import capturable_process_group, ddp_on_side_stream
                want_graph = capturable_process_group()      # gloo: the step stays eager
            if want_graph:
                # wrapper construction, every training / validation step and the capture share ONE side stream (gdlhip.graphs)
                model.model = ddp_on_side_stream(model.model, **ddp_kw)
                self._stream = model.model.gdl_stream
            else:
                model.model = nn.parallel.DistributedDataParallel(model.model, **ddp_kw)
        try:
            per_epoch = len(train_loader)
            self.estimated_stepping_batches = math.ceil(per_epoch / self.accumulate_grad_batches) * self.max_epochs
        except TypeError:
            self.estimated_stepping_batches = -1         # iterable loaders (the WebDataset datamodule)
        optimizers, sched_cfgs = model.configure_optimizers()
        opt = optimizers[0]
        self._graph_ok = want_graph
        step_opt = self._maybe_fuse(opt, device, capturable=self._graph_ok)
        self._graph_ok = self._graph_ok and getattr(step_opt, "capturable", False)
        self._graphed = None
        self._optimizers, self._sched_cfgs = [step_opt], sched_cfgs       # saved with every checkpoint
        best, bad_epochs = None, 0
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            self._run_train_epoch(model, train_loader, step_opt, opt, sched_cfgs, device)
            metrics = self._epoch_means(device)
            if hasattr(model, "on_train_epoch_end"):
                model.on_train_epoch_end()
            if val_loader is not None:
                metrics.update(self._run_eval(model, val_loader, "val", device))
                if hasattr(model, "on_validation_epoch_end"):
                    model.on_validation_epoch_end()
            self.callback_metrics.update(metrics)
            for cfg in sched_cfgs:
                if cfg.get("interval", "epoch") == "epoch" and (epoch + 1) % int(cfg.get("frequency", 1)) == 0:
                    self._step_scheduler(cfg, metrics)
            score = metrics.get(self.monitor)
            if score is not None:
                better = best is None or (score < best if self.mode == "min" else score > best)
                if better:
                    best, bad_epochs = score, 0
                    if self.is_global_zero:
                        name = self.checkpoint_filename.format(epoch=epoch, **{self.monitor: score}) + ".ckpt"
                        path = self.default_root_dir / "checkpoints" / name
                        old = self.checkpoint_callback.best_model_path
                        self.save_checkpoint(model, path)
                        if old and Path(old).exists() and Path(old) != path:
                            Path(old).unlink()           # save_top_k: 1
                        self.checkpoint_callback.best_model_path = str(path)
                    self.checkpoint_callback.best_model_score = score
                else:
                    bad_epochs += 1
            if self.is_global_zero:
                logger.info("epoch %d: %s", epoch, {k: round(v, 5) for k, v in metrics.items()})
            if self.early_stopping_patience is not None and bad_epochs >= self.early_stopping_patience:   # Lightning: wait_count >= patience
                break
        if self.world_size > 1:
            paths = [self.checkpoint_callback.best_model_path]
            dist.broadcast_object_list(paths, src=0)
            self.checkpoint_callback.best_model_path = paths[0]
            dist.barrier()

    def _maybe_fuse(self, opt: torch.optim.Optimizer, device: torch.device, capturable: bool = False):
        """torch.optim.Adam -> the fused multi-tensor Adam + clip kernels on the same param groups."""
        if not (self.use_fused_adam and device.type == "cuda" and type(opt) is torch.optim.Adam):
            return opt
        g0 = opt.param_groups[0]
        if any(g.get("amsgrad") or g.get("maximize") or g.get("capturable") for g in opt.param_groups):
            return opt
        from gdlhip.nn import FusedAdam
        fused = FusedAdam(opt.param_groups, lr=g0["lr"], betas=g0["betas"], eps=g0["eps"],
                          weight_decay=g0["weight_decay"], max_grad_norm=self.gradient_clip_val, capturable=capturable)
        fused.param_groups = opt.param_groups            # the scheduler keeps writing `lr` into these dicts
        return fused

    def _step_scheduler(self, cfg: dict[str, Any], metrics: dict[str, float]) -> None:
        sched = cfg["scheduler"]
        if isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau):
            key = cfg.get("monitor", self.monitor)
            if key not in metrics:
                msg = f"ReduceLROnPlateau monitors {key!r}, which was not logged (have {sorted(metrics)})"
                raise KeyError(msg)
            sched.step(metrics[key])
        else:
            sched.step()

    def _batches(self, loader, split: str):
        limit = self.limit[split]
        for i, batch in enumerate(loader):
            if limit is not None and i >= limit:
                break
            yield i, batch

    @contextlib.contextmanager
    def _on_stream(self):
        """Run a phase on the trainer's stream (the DDP wrapper's, when a captured DDP step is planned), ordered after what the
        default stream did before and before what it does next."""
        st = getattr(self, "_stream", None)
        if st is None:
            yield
            return
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            yield
        torch.cuda.current_stream().wait_stream(st)

    def _run_train_epoch(self, model, loader, step_opt, opt, sched_cfgs, device) -> None:
        with self._on_stream():
            self._train_epoch(model, loader, step_opt, opt, sched_cfgs, device)

    def _train_epoch(self, model, loader, step_opt, opt, sched_cfgs, device) -> None:
        model.train()
        self.training = True
        fused_clip = step_opt is not opt

        def optimizer_step() -> None:
            if self.gradient_clip_val and not fused_clip:
                torch.nn.utils.clip_grad_norm_(model.parameters(), self.gradient_clip_val)
            step_opt.step()
            step_opt.zero_grad(set_to_none=True)
            self.global_step += 1
            for cfg in sched_cfgs:
                if cfg.get("interval") == "step" and self.global_step % int(cfg.get("frequency", 1)) == 0:
                    self._step_scheduler(cfg, {})

        def after_step() -> None:
            self.global_step += 1
            for cfg in sched_cfgs:
                if cfg.get("interval") == "step" and self.global_step % int(cfg.get("frequency", 1)) == 0:
                    self._step_scheduler(cfg, {})

        step_opt.zero_grad(set_to_none=True)
        pending = False
        for i, batch in self._batches(loader, "train"):
            batch = model.on_before_batch_transfer(batch, 0) if hasattr(model, "on_before_batch_transfer") else batch
            batch = _to_device(batch, device)
            batch = model.on_after_batch_transfer(batch, 0) if hasattr(model, "on_after_batch_transfer") else batch
            if self._graph_ok and self._graph_step(model, step_opt, batch, device):
                after_step()
                continue
            if self._graphed is not None:
                # an eager step between replays (ragged batch): `p.grad` still names the graph's static gradient buffers, which
                # hold the last replay's gradients -- backward would ADD to them
                step_opt.zero_grad(set_to_none=True)
            with self._autocast(device), rng("forward+loss"):
                loss = model.training_step(batch, i)
            with rng("backward"):
                (loss / self.accumulate_grad_batches).backward()
            pending = True
            if (i + 1) % self.accumulate_grad_batches == 0:
                optimizer_step()
                pending = False
        if pending:              # Lightning steps on the last batch of an epoch even when the accumulation window is not full
            optimizer_step()
        self.training = False

    def _agree(self, tag: str, verdict: str, timeout_s: float = 120.0) -> str:
        """All ranks' verdicts on one question, through the process group's key-value store and NOT through a collective: the
        questions are asked exactly when ranks may be out of step with each other (one failed before the warm-up collectives, one
        died inside a capture) -- a 1-element all-reduce issued by the failing rank would pair up with a peer's DDP bucket
        all-reduce.  Returns the worst verdict ("fatal" < "no" < "ok"); a rank that does not answer within the timeout counts as
        "fatal".  Every call uses a fresh key generation."""
        if not (getattr(self, "_ddp_active", False) and self.world_size > 1 and dist.is_available() and dist.is_initialized()):
            return verdict
        from datetime import timedelta
        self._agree_gen = getattr(self, "_agree_gen", 0) + 1
        store = dist.distributed_c10d._get_default_store()
        base = f"gdl_agree/{tag}/{self._agree_gen}"
        store.set(f"{base}/{self.global_rank}", verdict)
        keys = [f"{base}/{r}" for r in range(self.world_size)]
        try:
            store.wait(keys, timedelta(seconds=timeout_s))
            got = [store.get(k).decode() for k in keys]
        except Exception as exc:  # noqa: BLE001  (timeout: a peer is gone)
            logger.error("MiniTrainer: no answer from every rank on '%s' within %.0f s (%s)", tag, timeout_s, exc)
            return "fatal"
        order = {"fatal": 0, "no": 1, "ok": 2}
        return min(got, key=lambda g: order.get(g, 0))

    def _graph_step(self, model, step_opt, batch, device) -> bool:
        """One training step as a hipGraph replay; False = this batch has to run eagerly (no graph yet and the batch is too
        large for "auto", another shape than the captured one, or the capture failed)."""
        from gdlhip.graphs import GraphCaptureFatal, GraphedTrainStep
        tensors = {k: v for k, v in batch.items() if isinstance(v, Tensor) and v.is_cuda} if isinstance(batch, dict) else {}
        multi = getattr(self, "_ddp_active", False) and self.world_size > 1
        if not tensors and not (multi and self._graphed is None):
            return False
        if self._graphed is None:
            lead = max((v.shape[0] for v in tensors.values() if v.dim() > 0), default=0)
            # rank-local reasons not to capture (no device tensors in the batch; "auto" and a GPU-bound batch size) are settled
            # with the peers BEFORE anybody constructs GraphedTrainStep: its eleven warm-up iterations are DDP collectives, and a
            # rank that skipped them would leave the others waiting in a bucket all-reduce (advisor, round 5)
            local_ok = bool(tensors) and not (self.graph_step == "auto" and lead > self.graph_max_batch)
            pre = self._agree("precondition", "ok" if local_ok else "no")
            if pre != "ok":
                if pre == "fatal":
                    raise RuntimeError("MiniTrainer: a rank did not answer before the hipGraph capture of the DDP step")
                self._graph_ok = False       # GPU-bound step (a graph buys nothing and doubles the activation memory), or a peer's veto
                return False
            amp = torch.bfloat16 if self.precision in ("bf16-mixed", "bf16", "16-mixed", "16") else None
            failure, fatal = None, None
            try:
                # (under DDP the warm-up is raised to the 11 eager iterations torch asks for; all of them are undone)
                self._graphed = GraphedTrainStep(model, step_opt, batch, autocast_dtype=amp, warmup=2, restore_state=True)
            except GraphCaptureFatal as exc:
                fatal = exc              # the device / RNG state of this process is gone; tell the peers BEFORE raising
            except Exception as exc:  # noqa: BLE001  (anything the capture cannot record: fall back to eager steps for good)
                failure, self._graphed = f"{type(exc).__name__}: {exc}", None
                logger.debug("capture traceback", exc_info=True)
                self.capture_traceback = traceback.format_exc()[-3000:]
            # every rank ran the same warm-up collectives and then recorded (not ran) the captured ones.  The outcome is agreed
            # through the store: all replay, or all run eagerly, or -- one rank's capture died with a HIP error -- all stop with
            # a message instead of waiting for that rank in the next collective until the watchdog fires
            outcome = self._agree("capture", "fatal" if fatal is not None else "no" if failure else "ok")
            if fatal is not None:
                raise fatal
            if outcome == "fatal":
                raise GraphCaptureFatal("the hipGraph capture of the DDP training step died on another rank (or that rank did not "
                                        "answer): stopping this rank too; rerun with graph_step=False")
            if outcome != "ok" and failure is None:
                failure, self._graphed = "the capture failed on another rank", None
            if failure is not None:
                logger.warning("MiniTrainer: hipGraph capture of the training step failed (%s); running eagerly", failure)
                self._graph_ok = False
                return False
            logger.info("MiniTrainer: training step captured into a hipGraph (per-GPU batch %d%s)", lead,
                        ", DDP collectives included" if getattr(self, "_ddp_active", False) else "")
        static = self._graphed.static
        for k, v in tensors.items():
            s = static.get(k)
            if not isinstance(s, Tensor) or s.shape != v.shape or s.dtype != v.dtype:
                return False                 # e.g. the ragged last batch of an epoch
        for k, v in batch.items():           # host-side entries (wavelengths) are baked into the captured step
            if isinstance(v, Tensor) and not v.is_cuda:
                s = static.get(k)
                if not isinstance(s, Tensor) or s.shape != v.shape or not torch.equal(s, v):
                    return False
        self._graphed(batch)
        self.graphed_steps += 1
        return True

    def _run_eval(self, model, loader, split: str, device) -> dict[str, float]:
        with self._on_stream():
            return self._eval_epoch(model, loader, split, device)

    @torch.no_grad()
    def _eval_epoch(self, model, loader, split: str, device) -> dict[str, float]:
        model.eval()
        self.training = False
        step = model.validation_step if split == "val" else model.test_step
        for i, batch in self._batches(loader, split):
            batch = model.on_before_batch_transfer(batch, 0) if hasattr(model, "on_before_batch_transfer") else batch
            batch = _to_device(batch, device)
            batch = model.on_after_batch_transfer(batch, 0) if hasattr(model, "on_after_batch_transfer") else batch
            with self._autocast(device):
                step(batch, i)
        return self._epoch_means(device)

    # ------------------------------------------------------------------ validate / test
    def validate(self, model, dataloaders=None, datamodule=None) -> list[dict[str, float]]:
        return [self._eval_entry(model, dataloaders, datamodule, "val")]

    def test(self, model, dataloaders=None, datamodule=None, ckpt_path: str | None = None) -> list[dict[str, float]]:
        if ckpt_path:
            model.trainer = self
            model.configure_model()
            self.load_checkpoint(model, ckpt_path)
        return [self._eval_entry(model, dataloaders, datamodule, "test")]

    def _eval_entry(self, model, loader, datamodule, split: str) -> dict[str, float]:
        device = self._device()
        model.trainer = self
        model.configure_model()
        model.to(device)
        if loader is None and datamodule is not None:
            datamodule.setup("test" if split == "test" else "validate")
            loader = datamodule.test_dataloader() if split == "test" else datamodule.val_dataloader()
        metrics = self._run_eval(model, loader, split, device)
        hook = getattr(model, "on_test_epoch_end" if split == "test" else "on_validation_epoch_end", None)
        if hook is not None:
            hook()
        self.callback_metrics.update(metrics)
        return metrics
