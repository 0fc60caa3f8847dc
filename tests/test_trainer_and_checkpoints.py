"""MiniTrainer / train.py (the reference's ``train.py fit`` flow without Lightning, train.py:27-80) and the checkpoint
formats either side of it (utils/models.py:10-66, dofa_v2.py:286-392) -- CPU tests on a toy task with the task classes'
hook surface.  The GPU versions (real DOFA task through the same trainer) are in tests/test_hip_tasks.py."""

import os
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

from gdlhip.trainer import MiniTrainer, seed_everything
from geo_deep_learning import train as gdl_train
from geo_deep_learning.utils.models import load_weights_from_checkpoint

import _toy_task as toy

CONFIG = {
    "seed_everything": True,
    "trainer": {"max_epochs": 4, "gradient_clip_val": 1.0, "precision": "16-mixed", "sync_batchnorm": True,
                "accelerator": "gpu", "devices": -1,
                "strategy": {"class_path": "lightning.pytorch.strategies.DDPStrategy", "init_args": {"find_unused_parameters": False}},
                "logger": {"class_path": "lightning.pytorch.loggers.mlflow.MLFlowLogger", "init_args": {"save_dir": "/nowhere"}},
                "callbacks": [{"class_path": "lightning.pytorch.callbacks.EarlyStopping", "init_args": {"monitor": "val_loss", "patience": 20}},
                              {"class_path": "lightning.pytorch.callbacks.ModelCheckpoint",
                               "init_args": {"monitor": "val_loss", "mode": "min", "save_top_k": 1, "filename": "model-{epoch:02d}-{val_loss:.3f}"}},
                              {"class_path": "tools.callbacks.segmentation_visualization.VisualizationCallback",
                               "init_args": {"num_classes": "${model.init_args.num_classes}"}}]},
    "model": {"class_path": "_toy_task.ToyTask",
              "init_args": {"num_classes": 3, "mean": "${data.init_args.batch_size}",
                            "loss": {"class_path": "torch.nn.CrossEntropyLoss"},
                            "optimizer": {"class_path": "torch.optim.Adam", "init_args": {"lr": 0.05}},
                            "scheduler": {"class_path": "torch.optim.lr_scheduler.ReduceLROnPlateau",
                                          "init_args": {"mode": "min", "factor": 0.1, "patience": 0, "min_lr": 1e-6}},
                            "scheduler_config": {"interval": "epoch", "frequency": 1, "monitor": "val_loss"}}},
    "data": {"class_path": "_toy_task.ToyData", "init_args": {"batch_size": 4, "num_classes": 3}},
    "ckpt_path": None,
}


def test_train_py_fit_checkpoints_and_post_fit_test(tmp_path):
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(CONFIG))
    out = gdl_train.main(["fit", "--config", str(cfg_path), f"--trainer.default_root_dir={tmp_path}", "--trainer.max_epochs=5"])
    best = Path(out["best_model_path"])
    assert best.is_file() and best.parent == tmp_path / "checkpoints"
    assert best.name.startswith("model-epoch=") is False and best.name.startswith("model-") and best.suffix == ".ckpt"
    assert len(list(best.parent.iterdir())) == 1                          # save_top_k: 1
    ckpt = torch.load(best)
    assert all(k.startswith("model.") for k in ckpt["state_dict"])        # Lightning layout (utils/models.py:33)
    assert {"model.encoder.weight", "model.norm.running_mean", "model.head.bias"} <= set(ckpt["state_dict"])
    assert ckpt["hyper_parameters"]["num_classes"] == 3 and ckpt["hyper_parameters"]["mean"] == 4   # ${...} resolved
    assert out["fit"]["val_loss"] < 1.0986 and "train_loss" in out["fit"]                            # learned something
    assert set(out["test"]) == {"test_loss"} and abs(out["test"]["test_loss"] - out["fit"]["val_loss"]) < 0.5


def test_minitrainer_hooks_scheduler_and_seed(tmp_path):
    seed_everything(42)
    a = torch.rand(3)
    seed_everything(42)
    assert torch.equal(a, torch.rand(3)) and os.environ["PL_GLOBAL_SEED"] == "42"
    task = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), optimizer=lambda p: torch.optim.SGD(p, lr=0.1),
                       scheduler=lambda o: torch.optim.lr_scheduler.ReduceLROnPlateau(o, factor=0.5, patience=0),
                       scheduler_config={"interval": "epoch", "monitor": "val_loss"})
    data = toy.ToyData(train_batches=3)
    tr = MiniTrainer(max_epochs=3, default_root_dir=str(tmp_path), gradient_clip_val=1.0, accumulate_grad_batches=1)
    tr.fit(task, datamodule=data)
    assert tr.global_step == 9 and tr.estimated_stepping_batches == 9
    assert task.calls.count("train_epoch_end") == 3 and task.calls.count("val_epoch_end") == 3
    assert task.calls.count("after:train") == 9 and task.calls.count("after:eval") == 6 and task.calls.count("before") == 15
    assert set(tr.callback_metrics) == {"train_loss", "val_loss"}
    # ReduceLROnPlateau is stepped with the monitored value: a scheduler monitoring a metric that is never logged fails
    bad = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), scheduler=lambda o: torch.optim.lr_scheduler.ReduceLROnPlateau(o),
                      scheduler_config={"interval": "epoch", "monitor": "val_iou"})
    with pytest.raises(KeyError, match="val_iou"):
        MiniTrainer(max_epochs=1, default_root_dir=str(tmp_path)).fit(bad, datamodule=toy.ToyData(train_batches=1))
    # accumulate_grad_batches: 6 batches -> 3 optimizer steps per epoch; interval "step" schedulers follow global_step
    acc = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), optimizer=lambda p: torch.optim.SGD(p, lr=1.0),
                      scheduler=lambda o: torch.optim.lr_scheduler.StepLR(o, 1, gamma=0.5),
                      scheduler_config={"interval": "step"})
    tr = MiniTrainer(max_epochs=1, default_root_dir=str(tmp_path), accumulate_grad_batches=2)
    tr.fit(acc, datamodule=toy.ToyData(train_batches=6))
    assert tr.global_step == 3


def test_checkpoint_round_trip_with_load_parts(tmp_path):
    """save -> load_weights_from_checkpoint(load_parts=...) (utils/models.py:10-66): prefix filter, model. stripping."""
    task = toy.ToyTask(3, torch.nn.CrossEntropyLoss())
    task.configure_model()
    tr = MiniTrainer(default_root_dir=str(tmp_path))
    path = tmp_path / "a.ckpt"
    tr.save_checkpoint(task, path)
    fresh = toy.ToyNet(3)
    res = load_weights_from_checkpoint(fresh, str(path), load_parts=["encoder", "norm"])
    assert sorted(res.missing_keys) == ["head.bias", "head.weight"] and not res.unexpected_keys
    assert torch.equal(fresh.encoder.weight, task.model.encoder.weight)
    assert not torch.equal(fresh.head.weight, task.model.head.weight)
    assert load_weights_from_checkpoint(fresh, str(path)) is None           # full, strict
    assert torch.equal(fresh.head.weight, task.model.head.weight)
    res = load_weights_from_checkpoint(toy.ToyNet(3), str(path), load_parts="nothing_here")
    assert len(res.missing_keys) == len(fresh.state_dict()) - 1             # all but num_batches_tracked (not "missing")
    torch.save(task.model.state_dict(), tmp_path / "bare.pth")              # a bare state dict (no "state_dict" key)
    assert load_weights_from_checkpoint(toy.ToyNet(3), str(tmp_path / "bare.pth")) is None


def test_dofa_pretrained_key_remap_and_pos_embed_resize(tmp_path, monkeypatch):
    """dofa_v2.py:286-392 on a synthetic DOFA-style checkpoint: ``model.``-prefixed ViT keys are kept, other ``model.*``
    keys dropped, ``patch_embed.*`` kept, a 16x16 position grid is resized bicubically to the model's grid (cls token
    untouched), missing / unexpected keys fail.  ``pretrained=True`` reads the torch-hub cache location."""
    from geo_deep_learning.models.encoders.dofa_v2 import DOFAv2
    kw = dict(img_size=84, patch_size=14, embed_dim=64, depth=2, num_heads=2, out_indices=[0, 1])
    enc = DOFAv2(pretrained=False, **kw)                                    # 6x6 grid: 37 tokens
    g = torch.Generator().manual_seed(0)
    src = {k: torch.randn(v.shape, generator=g) for k, v in enc.state_dict().items()}
    ck = {}
    for k, v in src.items():
        ck[k if k.startswith("patch_embed.") else f"model.{k}"] = v
    ck["model.pos_embed"] = torch.randn(1, 1 + 16 * 16, 64, generator=g)   # published checkpoints: 224 / 14 = 16
    ck["model.fc_norm.weight"] = torch.zeros(64)                            # dropped: not blocks./norm./cls/pos
    ck["model.head.weight"] = torch.zeros(10, 64)
    missing, unexpected = enc.load_pretrained_state_dict({"model": dict(ck)})
    assert not unexpected and set(missing) <= {"head.weight", "head.bias"}
    sd = enc.state_dict()
    for k in src:
        if k != "pos_embed":
            assert torch.equal(sd[k], src[k]), k
    grid = ck["model.pos_embed"][:, 1:].reshape(1, 16, 16, 64).permute(0, 3, 1, 2)
    want = torch.nn.functional.interpolate(grid, size=(6, 6), mode="bicubic", align_corners=False)
    assert torch.equal(sd["pos_embed"][:, 0], ck["model.pos_embed"][:, 0])
    assert torch.equal(sd["pos_embed"][:, 1:], want.permute(0, 2, 3, 1).reshape(1, 36, 64))
    bad = dict(ck)
    del bad["model.cls_token"]
    with pytest.raises(RuntimeError, match="Missing required keys"):
        DOFAv2(pretrained=False, **kw).load_pretrained_state_dict(bad)
    with pytest.raises(RuntimeError, match="Unexpected keys"):
        DOFAv2(pretrained=False, **kw).load_pretrained_state_dict({**ck, "patch_embed.extra": torch.zeros(1)})
    # pretrained=True: the file torch.hub.load_state_dict_from_url would have cached
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path))
    with pytest.raises(RuntimeError, match="dofav2_vit_base_e150.pth"):
        DOFAv2(encoder_name="dofa_base", pretrained=True, **kw)
    (tmp_path / "checkpoints").mkdir()
    torch.save({"model": ck}, tmp_path / "checkpoints" / "dofav2_vit_base_e150.pth")
    hot = DOFAv2(encoder_name="dofa_base", pretrained=True, **kw)
    assert torch.equal(hot.state_dict()["blocks.1.mlp.fc2.weight"], src["blocks.1.mlp.fc2.weight"])


def _ddp_worker(rank, world, port, tmp, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _toy_task as toy
        torch.manual_seed(0)
        task = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), optimizer=lambda p: torch.optim.SGD(p, lr=0.1), batchnorm=False)
        trn = toy.make_batches(4, 8, 3, 1, rank, world)
        val = toy.make_batches(2, 8, 3, 2, rank, world)
        tr = MiniTrainer(max_epochs=2, default_root_dir=tmp)     # (torch's SyncBatchNorm needs GPU modules: GPU test)
        tr.fit(task, train_dataloaders=trn, val_dataloaders=val)
        inner = task.model.module
        ret[rank] = {"w": inner.encoder.weight.detach().clone(), "val": tr.callback_metrics["val_loss"],
                     "best": tr.checkpoint_callback.best_model_path}
    finally:
        dist.destroy_process_group()


def _empty_val_rank_worker(rank, world, port, tmp, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _toy_task as toy
        torch.manual_seed(0)
        task = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), optimizer=lambda p: torch.optim.SGD(p, lr=0.1), batchnorm=False)
        trn = toy.make_batches(4, 8, 3, 1, rank, world)
        val = toy.make_batches(2, 8, 3, 2) if rank == 0 else []     # rank 1's slice of the validation shards is empty
        tr = MiniTrainer(max_epochs=2, default_root_dir=tmp)
        tr.fit(task, train_dataloaders=trn, val_dataloaders=val)
        ret[rank] = {"val": tr.callback_metrics.get("val_loss"), "best": tr.checkpoint_callback.best_model_path}
    finally:
        dist.destroy_process_group()


def test_minitrainer_rank_without_validation_batches_does_not_hang(tmp_path):
    """Round-3 advisor finding: a rank that logged nothing in validation skipped the epoch-mean all-reduce its peers were in
    (hang) and never saw the monitored metric.  The reduction is now shape-stable (union of the ranks' metric names, zeros for
    what a rank did not log): both ranks finish, hold the same val_loss (= rank 0's) and agree on the checkpoint."""
    world, port = 2, 33700 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_empty_val_rank_worker, args=(r, world, port, str(tmp_path), ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, "a rank hung or failed"
    assert ret[0]["val"] is not None and ret[1]["val"] is not None and abs(ret[0]["val"] - ret[1]["val"]) < 1e-12
    assert ret[0]["best"] == ret[1]["best"] and Path(ret[0]["best"]).is_file()


def test_minitrainer_ddp_world2_matches_single_process(tmp_path):
    """Two gloo ranks, each on half of every batch, under DDP == one process on the full batches (mean-reduced loss =>
    averaged gradients), metrics averaged over ranks, one checkpoint written by rank 0 and known to both."""
    world, port = 2, 29700 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, str(tmp_path), ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    torch.manual_seed(0)
    task = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), optimizer=lambda p: torch.optim.SGD(p, lr=0.1), batchnorm=False)
    tr = MiniTrainer(max_epochs=2, default_root_dir=str(tmp_path / "single"))
    tr.fit(task, train_dataloaders=toy.make_batches(4, 8, 3, 1), val_dataloaders=toy.make_batches(2, 8, 3, 2))
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["w"], r1["w"]) and r0["best"] == r1["best"] and Path(r0["best"]).is_file()
    assert torch.allclose(r0["w"], task.model.encoder.weight, atol=1e-5)
    assert abs(r0["val"] - tr.callback_metrics["val_loss"]) < 1e-5 and abs(r0["val"] - r1["val"]) < 1e-12
    sd = torch.load(r0["best"])["state_dict"]
    assert "model.encoder.weight" in sd and not any("module." in k for k in sd)


def _torchrun_worker(rank, world, port, cfg_path, tmp, ret):
    # what torch.distributed.run exports for every rank; the process group is NOT created here: train.main must do it
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as d
    seen = {}
    orig_fit = MiniTrainer.fit

    def spy(self, *a, **kw):
        seen.update(world=self.world_size, rank=self.global_rank, initialised=d.is_initialized())
        return orig_fit(self, *a, **kw)
    MiniTrainer.fit = spy
    out = gdl_train.main(["fit", "--config", cfg_path, f"--trainer.default_root_dir={tmp}", "--trainer.max_epochs=2",
                          "--trainer.sync_batchnorm=false", "--model.init_args.batchnorm=false"])
    ret[rank] = {"seen": seen, "best": out["best_model_path"], "tested": "test" in out, "alive": d.is_initialized()}


def test_train_py_under_torchrun_initialises_the_process_group(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m geo_deep_learning.train fit ...`: with only the
    launcher's environment set, train.main creates the process group itself (gloo here, nccl = RCCL on GPUs), the trainer
    sees world size 2, ONE checkpoint is written (by rank 0) and known to both ranks, only rank 0 runs the post-fit test,
    and the group is destroyed again at exit."""
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(CONFIG))
    world, port = 2, 31700 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_torchrun_worker, args=(r, world, port, str(cfg_path), str(tmp_path), ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    for r in range(world):
        assert ret[r]["seen"] == {"world": 2, "rank": r, "initialised": True} and not ret[r]["alive"]
    assert ret[0]["best"] == ret[1]["best"] and Path(ret[0]["best"]).is_file()
    assert len(list((tmp_path / "checkpoints").iterdir())) == 1
    assert ret[0]["tested"] and not ret[1]["tested"]
    ck = torch.load(ret[0]["best"])
    assert ck["pytorch-lightning_version"] and len(ck["optimizer_states"]) == 1 and len(ck["lr_schedulers"]) == 1


def test_accumulation_tail_early_stopping_and_metric_sink(tmp_path):
    """Lightning semantics MiniTrainer has to reproduce: an epoch whose batch count is not a multiple of
    accumulate_grad_batches still steps on its last batch (estimated_stepping_batches uses ceil); EarlyStopping stops when
    the number of epochs without improvement REACHES patience; logged tensors are reduced without a per-step host sync."""
    acc = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), optimizer=lambda p: torch.optim.SGD(p, lr=0.1))
    tr = MiniTrainer(max_epochs=2, default_root_dir=str(tmp_path), accumulate_grad_batches=2)
    tr.fit(acc, datamodule=toy.ToyData(train_batches=5))
    assert tr.global_step == 6 == tr.estimated_stepping_batches            # ceil(5 / 2) * 2
    # learning rate 0: val_loss never improves after epoch 0 -> with patience 1 training ends after epoch 1
    bad = toy.ToyTask(3, torch.nn.CrossEntropyLoss(), optimizer=lambda p: torch.optim.SGD(p, lr=0.0), batchnorm=False)
    tr = MiniTrainer(max_epochs=6, default_root_dir=str(tmp_path / "es"), early_stopping_patience=1)
    tr.fit(bad, datamodule=toy.ToyData(train_batches=2))
    assert tr.current_epoch == 1
    # the sink keeps tensors as tensors until the epoch mean is formed
    tr = MiniTrainer(default_root_dir=str(tmp_path))
    tr._collect("x", torch.tensor(2.0), 3)
    tr._collect("x", torch.tensor(4.0), 1)
    tr._collect("y", 1.5, None)
    assert isinstance(tr._sums["x"][0], torch.Tensor)
    means = tr._epoch_means(torch.device("cpu"))
    assert abs(means["x"] - 2.5) < 1e-12 and abs(means["y"] - 1.5) < 1e-12


# ------------------------------------------------------------------------------------------------ MiT ImageNet weights
# the model section of the reference's shipped SegFormer config (configs/segformer_config_RGB.yaml:41-66), restated as data
SEGFORMER_MODEL_SECTION = {
    "class_path": "tasks_with_models.segmentation_segformer.SegmentationSegformer",
    "init_args": {"encoder": "mit_b0", "image_size": [512, 512], "in_channels": 3, "weights": "imagenet", "max_samples": 6,
                  "num_classes": 5, "freeze_layers": None, "use_dynamic_encoder": False,
                  "loss": {"class_path": "segmentation_models_pytorch.losses.DiceLoss", "init_args": {"mode": "multiclass"}},
                  "optimizer": {"class_path": "torch.optim.Adam", "init_args": {"lr": 6e-5}},
                  "scheduler": {"class_path": "torch.optim.lr_scheduler.ReduceLROnPlateau",
                                "init_args": {"mode": "min", "factor": 0.5, "patience": 3}}}}


def test_mit_imagenet_weights_come_from_the_hub_cache(tmp_path, monkeypatch):
    """`weights: imagenet` of configs/segformer_config_RGB.yaml:46.  The reference downloads
    .../segmentation_models.pytorch/releases/download/v0.0.2/<name>.pth into torch-hub's cache with model_zoo.load_url and drops
    the checkpoint's classifier `head.*` (mix_transformer.py:580-596,732-746).  This build reads the same cache location (or
    $GDL_MIT_CHECKPOINT), never downloads, and fails with the path it wants when the file is not there."""
    from geo_deep_learning.models.encoders import mix_transformer as mit
    from oracle.segformer import SegFormerSegmentationModel as OracleSegFormer
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path / "hub"))
    monkeypatch.delenv("GDL_MIT_CHECKPOINT", raising=False)
    # no file: loud, with the path
    with pytest.raises(RuntimeError, match="mit_b0.pth") as err:
        mit.get_encoder("mit_b0", weights="imagenet")
    assert str(tmp_path / "hub" / "checkpoints" / "mit_b0.pth") in str(err.value) and "GDL_MIT_CHECKPOINT" in str(err.value)
    with pytest.raises(KeyError, match="Wrong pretrained weights"):
        mit.get_encoder("mit_b0", weights="ssl")
    with pytest.raises(KeyError, match="Wrong encoder name"):
        mit.get_encoder("mit_b9", weights="imagenet")
    with pytest.warns(UserWarning, match="non-RGB"):      # the reference warns and keeps the random init (:747-752)
        mit.get_encoder("mit_b0", in_channels=4, weights="imagenet")
    with pytest.raises(ValueError, match="dilated"):
        mit.get_encoder("mit_b0", output_stride=16)
    # a checkpoint in the published layout: the encoder's own keys + the ImageNet classifier head
    torch.manual_seed(3)
    src = OracleSegFormer("mit_b0", 3, 5).encoder.state_dict()
    ckpt = {k: torch.randn_like(v) if v.is_floating_point() else v.clone() for k, v in src.items()}
    ckpt["head.weight"], ckpt["head.bias"] = torch.randn(1000, 256), torch.randn(1000)
    (tmp_path / "hub" / "checkpoints").mkdir(parents=True)
    torch.save(ckpt, tmp_path / "hub" / "checkpoints" / "mit_b0.pth")
    enc = mit.get_encoder("mit_b0", weights="imagenet")
    sd = enc.state_dict()
    assert sorted(sd) == sorted(k for k in ckpt if not k.startswith("head."))
    assert all(torch.equal(sd[k], ckpt[k]) for k in sd)
    # ... the environment variable wins (file or directory), and the reference's shipped model section builds unchanged
    other = tmp_path / "elsewhere"
    other.mkdir()
    ckpt2 = {k: (v + 1 if v.is_floating_point() else v) for k, v in ckpt.items()}
    torch.save(ckpt2, other / "mit_b0.pth")
    monkeypatch.setenv("GDL_MIT_CHECKPOINT", str(other))
    task = gdl_train.instantiate(SEGFORMER_MODEL_SECTION)
    task.configure_model()
    got = task.model.encoder.state_dict()
    assert all(torch.equal(got[k], ckpt2[k]) for k in got)
    monkeypatch.setenv("GDL_MIT_CHECKPOINT", str(other / "mit_b0.pth"))
    assert torch.equal(mit.get_encoder("mit_b0", weights="imagenet").state_dict()["norm4.weight"], ckpt2["norm4.weight"])
    # use_dynamic_encoder=True takes stages 2-4 / blocks / norms from the same checkpoint (mix_transformer.py:862-895)
    dyn = mit.DynamicMixTransformer("mit_b0", weights="imagenet")
    assert torch.equal(dyn.state_dict()["block1.0.attn.q.weight"], ckpt2["block1.0.attn.q.weight"])
