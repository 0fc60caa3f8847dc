"""``python -m geo_deep_learning.train fit --config <yaml>`` without Lightning.

Mirror of the reference's entry point (/root/reference/geo_deep_learning/train.py:27-80): seed 42, build model /
datamodule / trainer from the SAME yaml files (``class_path`` + ``init_args``, ``${a.b}`` interpolation, dotted
``--a.b=value`` overrides), ``fit``, then -- like ``GeoDeepLearningCLI.after_fit`` (:30-63) -- test the best checkpoint on
rank 0.  When ``lightning`` is importable the reference's own ``train.py`` is the better entry point and this file is not
needed; here ``gdlhip.trainer.MiniTrainer`` drives the same hooks.

What is read from the yaml: ``model`` (any of the three task classes), ``data`` (``MultiSensorDataModule``), and from
``trainer``: max_epochs, precision, gradient_clip_val, sync_batchnorm, accumulate_grad_batches, limit_*_batches,
default_root_dir plus the ``ModelCheckpoint`` / ``EarlyStopping`` entries of ``callbacks`` (monitor, mode, filename,
patience).  MLflow logger, visualisation callbacks and strategy objects are control plane outside the hot path (SURVEY.md
section 2) and are skipped with a log line.  Third-party classes that are absent map to the build's own:
``segmentation_models_pytorch.losses.DiceLoss`` -> ``gdlhip.nn.DiceLoss``.
"""

from __future__ import annotations

import copy
import functools
import importlib
import logging
import os
import re
import sys
from pathlib import Path
from typing import Any

import yaml

from gdlhip.trainer import MiniTrainer, seed_everything

logger = logging.getLogger(__name__)

CLASS_ALIASES = {"segmentation_models_pytorch.losses.DiceLoss": "gdlhip.nn.DiceLoss"}
CALLABLE_KEYS = ("optimizer", "scheduler")       # LightningCLI OptimizerCallable / LRSchedulerCallable arguments
TRAINER_KEYS = ("max_epochs", "precision", "gradient_clip_val", "sync_batchnorm", "accumulate_grad_batches",
                "limit_train_batches", "limit_val_batches", "limit_test_batches", "default_root_dir", "fast_dev_run")


def _lookup(cfg: dict[str, Any], dotted: str) -> Any:
    cur: Any = cfg
    for part in dotted.split("."):
        cur = cur[int(part)] if isinstance(cur, list) else cur[part]
    return cur


def resolve_interpolations(cfg: dict[str, Any]) -> dict[str, Any]:
    """omegaconf-style ``${data.init_args.mean}`` references (configs/dofa_config_RGB.yaml:37-41,52-54)."""
    pat = re.compile(r"^\$\{([^}]+)\}$")

    def walk(node: Any) -> Any:
        if isinstance(node, dict):
            return {k: walk(v) for k, v in node.items()}
        if isinstance(node, list):
            return [walk(v) for v in node]
        if isinstance(node, str):
            m = pat.match(node)
            if m:
                return walk(_lookup(cfg, m.group(1)))
        return node

    return walk(cfg)


def apply_overrides(cfg: dict[str, Any], overrides: list[str]) -> None:
    """``--trainer.max_epochs=2`` / ``--model.init_args.num_classes 5`` style overrides (values parsed as yaml)."""
    i = 0
    while i < len(overrides):
        tok = overrides[i]
        if not tok.startswith("--"):
            msg = f"unexpected argument {tok!r}"
            raise SystemExit(msg)
        key, _, val = tok[2:].partition("=")
        if not _:
            i += 1
            val = overrides[i]
        cur = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = yaml.safe_load(val)
        i += 1


def _import(class_path: str) -> Any:
    class_path = CLASS_ALIASES.get(class_path, class_path)
    mod, _, name = class_path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node: Any) -> Any:
    """``{class_path, init_args}`` -> object; nested specs are instantiated first, ``optimizer`` / ``scheduler`` specs
    become ``functools.partial(cls, **init_args)`` (called later with the parameters / the optimizer)."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "class_path" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    kwargs = {}
    for k, v in (node.get("init_args") or {}).items():
        if k in CALLABLE_KEYS and isinstance(v, dict) and "class_path" in v:
            kwargs[k] = functools.partial(_import(v["class_path"]), **(v.get("init_args") or {}))
        else:
            kwargs[k] = instantiate(v)
    return _import(node["class_path"])(**kwargs)


def build(cfg: dict[str, Any]):
    """(model, datamodule, trainer) from a resolved config dict."""
    model = instantiate(cfg["model"])
    sched = (cfg["model"].get("init_args") or {}).get("scheduler")
    if isinstance(sched, dict):
        model.hparams["scheduler"] = sched       # what LightningCLI stores: configure_optimizers reads its class_path
    datamodule = instantiate(cfg["data"]) if cfg.get("data") else None
    tcfg = dict(cfg.get("trainer") or {})
    kw = {k: tcfg[k] for k in TRAINER_KEYS if k in tcfg}
    for cb in tcfg.get("callbacks") or []:
        name, init = cb.get("class_path", "").rsplit(".", 1)[-1], cb.get("init_args") or {}
        if name == "ModelCheckpoint":
            kw.update(monitor=init.get("monitor", "val_loss"), mode=init.get("mode", "min"))
            if init.get("filename"):
                kw["checkpoint_filename"] = init["filename"]
        elif name == "EarlyStopping":
            kw["early_stopping_patience"] = init.get("patience", 3)
        else:
            logger.info("train.py: callback %s is outside the hot path; skipped", cb.get("class_path"))
    for skipped in ("logger", "strategy"):
        if skipped in tcfg:
            logger.info("train.py: trainer.%s is handled by MiniTrainer itself (DDP over torch.distributed); skipped", skipped)
    return model, datamodule, MiniTrainer(**kw)


def init_distributed() -> bool:
    """One process per GPU under ``torchrun`` / ``torch.distributed.run``: when the launcher's environment names a world of
    more than one rank and no process group exists yet, bind this rank to its GPU and create the default group (RCCL =
    backend "nccl" on GPUs, "gloo" otherwise) -- what Lightning's DDP strategy does for the reference
    (configs/dofa_config_RGB.yaml:3-13).  Returns True when this call created the group (the caller destroys it)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or dist.is_initialized() or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    if "RANK" not in os.environ:
        msg = "WORLD_SIZE > 1 but RANK is not set: launch with torch.distributed.run (one process per GPU)"
        raise RuntimeError(msg)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        # the whole-step capture under DDP (gdlhip.graphs) needs the process group's asynchronous error handling off -- its
        # watchdog polls events of earlier collectives while a capture records -- and an eagerly bound device
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    return True


def main(args: list[str] | None = None) -> dict[str, Any]:
    """Run ``fit`` (+ the post-fit test of the best checkpoint).  Returns the trainer's metrics (for tests)."""
    created_group = init_distributed()
    if not created_group:
        return _main(args)
    import torch.distributed as dist
    try:
        out = _main(args)
        # after_fit: `self.trainer.strategy.barrier()` -- the other ranks wait for rank 0's post-fit test.  Only on the success
        # path: a rank that raised must not park its peers in a collective until the watchdog fires (and mask its own error)
        dist.barrier()
        return out
    finally:
        dist.destroy_process_group()


def _main(args: list[str] | None = None) -> dict[str, Any]:
    args = list(sys.argv[1:] if args is None else args)
    if not args or args[0] not in ("fit", "validate", "test"):
        msg = "usage: train.py {fit|validate|test} --config <yaml> [--dotted.key=value ...]"
        raise SystemExit(msg)
    command, rest = args[0], args[1:]
    cfg: dict[str, Any] = {}
    overrides = []
    i = 0
    while i < len(rest):
        if rest[i] in ("--config", "-c"):
            with Path(rest[i + 1]).open() as f:
                cfg.update(yaml.safe_load(f) or {})
            i += 2
        elif rest[i].startswith("--config="):
            with Path(rest[i].split("=", 1)[1]).open() as f:
                cfg.update(yaml.safe_load(f) or {})
            i += 1
        else:
            overrides.append(rest[i])
            i += 1
    apply_overrides(cfg, overrides)
    cfg = resolve_interpolations(cfg)
    seed_everything(42, workers=True)             # train.py:67
    model, datamodule, trainer = build(cfg)
    ckpt_path = cfg.get("ckpt_path")
    out: dict[str, Any] = {}
    if command == "fit":
        trainer.fit(model, datamodule=datamodule)
        out["fit"] = dict(trainer.callback_metrics)
        # GeoDeepLearningCLI.after_fit (train.py:30-63): rank 0 tests the best checkpoint on the test split
        test_loader = datamodule.test_dataloader() if datamodule is not None else None
        best = trainer.checkpoint_callback.best_model_path
        if trainer.is_global_zero and test_loader is not None and best:
            tester = MiniTrainer(precision=trainer.precision, default_root_dir=str(trainer.default_root_dir),
                                 limit_test_batches=trainer.limit["test"])
            tester.world_size, tester.global_rank = 1, 0
            # like load_from_checkpoint(..., weights_from_checkpoint_path=None) in after_fit (train.py:52-56): the tested
            # weights are the best checkpoint's, not the fine-tuning start point
            model_cfg = copy.deepcopy(cfg["model"])
            if "weights_from_checkpoint_path" in (model_cfg.get("init_args") or {}):
                model_cfg["init_args"]["weights_from_checkpoint_path"] = None
            fresh = instantiate(model_cfg)
            out["test"] = tester.test(fresh, dataloaders=test_loader, ckpt_path=best)[0]
            logger.info("Test metrics of %s: %s", best, out["test"])
        elif trainer.is_global_zero and test_loader is None:
            logger.warning("No test dataloader found.")
        out["best_model_path"] = best
    elif command == "validate":
        if ckpt_path:
            model.trainer = trainer
            model.configure_model()
            trainer.load_checkpoint(model, ckpt_path)
        out["validate"] = trainer.validate(model, datamodule=datamodule)[0]
    else:
        out["test"] = trainer.test(model, datamodule=datamodule, ckpt_path=ckpt_path)[0]
    if trainer.is_global_zero:
        logger.info("Done!")
    return out


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main()
