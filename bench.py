#!/usr/bin/env python
"""Headline benchmark: 512x512 tiles/s, DOFA-base + UperNet, bf16, on N MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch that is already resident in HBM:
  train  = SegmentationDOFA.training_step (forward, Dice main + 0.4 aux) + backward + global-norm
           clip 1.0 + Adam(6e-5), encoder frozen (configs/dofa_config_RGB.yaml:11,57-65), bf16
           autocast, per-GPU batch fixed (weak scaling), DDP over RCCL with SyncBatchNorm;
  infer  = forward + softmax->argmax mask under no_grad.
`value` is the TRAINING throughput of the whole job (tiles/s over all ranks); the inference
throughput measured the same way is reported alongside.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "geo-deep-learning_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
UNETPP_R18_FWD_GF = 128.04  # sum over convs of 2*K*C*R*S*Hout*Wout at 512x512 (counted on oracle/unetpp.py)
RGB_MEAN = [0.3992, 0.4283, 0.3998]   # configs/dofa_config_RGB.yaml:91-98
RGB_STD = [0.1672, 0.1800, 0.1584]
WAVELENGTHS = [0.665, 0.549, 0.481]   # configs/dofa_config_RGB.yaml:50


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mode", default="both", choices=["both", "train", "infer"])
    ap.add_argument("--model", default="dofa", choices=["dofa", "segformer", "unetpp"],
                    help="dofa = DOFA-base+UperNet (headline, configs[1]); segformer = SegFormer-B2 (configs[2], all parameters trainable); "
                         "unetpp = UNet++/ResNet18 (configs[0], the reference's CPU smoke case)")
    ap.add_argument("--with-input-stage", action="store_true",
                    help="also time the train step fed by host uint8 tiles through DeviceInputStage (PCIe-inclusive; "
                         "reported beside `value`, never as `value`)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="take the multi-GPU code path (RCCL group, SyncBatchNorm, DDP) even with one rank (self-test)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    return ap.parse_args()


def synthetic_batch(batch: int, device, seed: int):
    """uint8 U{0..255} tile -> /255 -> standardise (HIP kernel); mask U{0..4}."""
    from geo_deep_learning.utils.tensors import normalize_standardize_u8
    g = torch.Generator(device="cpu").manual_seed(seed)
    u8 = torch.randint(0, 256, (batch, 3, 512, 512), generator=g, dtype=torch.uint8).to(device)
    mask = torch.randint(0, 5, (batch, 1, 512, 512), generator=g, dtype=torch.int64).to(device)
    image = normalize_standardize_u8(u8, torch.tensor(RGB_MEAN, device=device), torch.tensor(RGB_STD, device=device))
    return {"image": image, "mask": mask, "wavelengths": torch.tensor(WAVELENGTHS)}


def timed(fn, steps: int, warmup: int, world: int, device) -> float:
    """W untimed + exactly K timed steps, barrier + synchronize on both sides, max over ranks."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt


def cpu_baseline():
    """The oracle (CPU restatement of the reference path, validated against reference goldens)
    timed on this box's host cores: ONE train step + ONE eval forward at batch 2, f32."""
    import oracle
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    m = oracle.DOFASegmentationModel("dofa_base", (512, 512), num_classes=5, freeze_layers=["encoder"])
    b = oracle.synthetic_batch(2, 3, 512, 5, 42)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=6e-5)
    m.train()
    t0 = time.perf_counter()
    loss = oracle.model.training_loss(m(b["image"], b["wavelengths"]), b["mask"])
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    t_train = time.perf_counter() - t0
    m.eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        oracle.model.predict_mask(m(b["image"], b["wavelengths"]))
    t_inf = time.perf_counter() - t0
    return {"value": round(2 / t_train, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": f"1 train step (fwd+bwd+clip+Adam) at batch 2, f32, {t_train:.1f}s; "
                      f"inference 1 forward+argmax at batch 2: {2 / t_inf:.3f} tiles/s",
            "inference_value": round(2 / t_inf, 4)}


def main() -> None:
    args = parse()
    # RCCL prints a version banner to the C-level stdout of every rank: keep a private handle on the real stdout for
    # the ONE JSON line and send everything else written to fd 1 to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_on = world > 1 or args.force_ddp
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=device)

    from gdlhip import ops
    from gdlhip.nn import DiceLoss, FusedAdam
    from tasks_with_models.segmentation_dofa import SegmentationDOFA
    from tasks_with_models.segmentation_segformer import SegmentationSegformer

    torch.manual_seed(42 + rank)  # train.py:67 seeds 42
    if args.model == "unetpp":
        from tasks_with_models.segmentation_unetplus import SegmentationUnetPlus
        task = SegmentationUnetPlus(encoder="resnet18", image_size=(512, 512), in_channels=3, num_classes=5,
                                    max_samples=6, loss=DiceLoss(mode="multiclass"),
                                    optimizer=lambda params: FusedAdam(params, lr=6e-5, max_grad_norm=1.0))
    elif args.model == "segformer":
        task = SegmentationSegformer(encoder="mit_b2", in_channels=3, num_classes=5, max_samples=6,
                                     loss=DiceLoss(mode="multiclass"),
                                     optimizer=lambda params: FusedAdam(params, lr=6e-5, max_grad_norm=1.0))
    else:
        task = SegmentationDOFA(
            encoder="dofa_base", pretrained=False, image_size=(512, 512), num_classes=5, max_samples=6,
            loss=DiceLoss(mode="multiclass"), freeze_layers=["encoder"],
            optimizer=lambda params: FusedAdam(params, lr=6e-5, max_grad_norm=1.0))
    task.configure_model()
    task.to(device)
    if dist_on:
        # Lightning's `sync_batchnorm: true` + DDPStrategy(gradient_as_bucket_view=true)
        task.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(task.model)
        task.model = torch.nn.parallel.DistributedDataParallel(
            task.model, device_ids=[local], gradient_as_bucket_view=True, find_unused_parameters=False)
    (optimizer,), _ = task.configure_optimizers()
    batch = synthetic_batch(args.batch, device, 42 + rank)
    use_bf16 = args.dtype == "bf16"

    def train_step():
        task.train()
        optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
            loss = task.training_step(batch, 0)
        loss.backward()
        optimizer.step()

    def infer_step():
        task.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
            task.validation_step(batch, 0)

    timer = None
    res = {}
    if args.mode in ("both", "train"):
        if not args.no_kernel_timer:
            for _ in range(args.warmup):   # warm up untimed, then time WITH the kernel events on
                train_step()
            timer = ops.KernelTimer()
            ops.TIMER = timer
            dt = timed(train_step, args.steps, 0, world, device)
            ops.TIMER = None
        else:
            dt = timed(train_step, args.steps, args.warmup, world, device)
        res["train"] = dt
    if args.mode in ("both", "infer"):
        res["infer"] = timed(infer_step, args.steps, args.warmup, world, device)

    pcie = None
    if args.with_input_stage and "train" in res:
        # host batches exactly as the dataset workers hand them over: raw uint8 tiles + int64 masks + sensor stats
        from geo_deep_learning.datamodules.device_input import DeviceInputStage
        g = torch.Generator(device="cpu").manual_seed(7 + rank)
        host = [{"image": torch.randint(0, 256, (args.batch, 3, 512, 512), generator=g, dtype=torch.uint8),
                 "mask": torch.randint(0, 5, (args.batch, 1, 512, 512), generator=g, dtype=torch.int64),
                 "wavelengths": torch.tensor(WAVELENGTHS),
                 "mean": torch.tensor(RGB_MEAN).view(1, 3, 1, 1).expand(args.batch, 3, 1, 1).contiguous(),
                 "std": torch.tensor(RGB_STD).view(1, 3, 1, 1).expand(args.batch, 3, 1, 1).contiguous()}
                for _ in range(3)]
        n_total = args.warmup + args.steps
        stage = DeviceInputStage((host[i % 3] for i in range(n_total)), device, depth=2)
        it = iter(stage)
        resident = batch

        def staged_step():
            nonlocal batch
            batch = next(it)
            train_step()
        dt = timed(staged_step, args.steps, args.warmup, world, device)
        batch = resident
        pcie = {"train_tiles_per_s": round(args.batch * world * args.steps / dt, 3),
                "h2d_bytes_per_tile": stage.bytes_h2d // (n_total * args.batch),
                "note": "host uint8 tiles -> pinned ring -> copy stream (2 batches ahead) -> normalise kernel -> step"}

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    tiles = args.batch * world * args.steps
    head = "train" if "train" in res else "infer"
    model_name = {"segformer": "SegFormer-B2 (MiT-B2 + MLP decoder)", "unetpp": "UNet++ (ResNet18 encoder)",
                  "dofa": "DOFA-base + UperNet"}[args.model]
    cfg_name = {"segformer": "configs[2]", "unetpp": "configs[0]", "dofa": "configs[1]"}[args.model]
    out = {
        "metric": f"512x512 tiles/s, {model_name}, {head} step",
        "value": round(tiles / res[head], 3),
        "unit": "tiles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * res[head] / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": (f"{model_name}, 3-band RGB 512x512 ({cfg_name}): "
                         + (("training step = fwd + Dice + bwd (every parameter) + clip 1.0 + Adam 6e-5, "
                             "DropPath/Dropout2d active" if args.model != "dofa" else
                             "training step = fwd + Dice(main)+0.4*Dice(aux) + bwd + clip 1.0 + Adam 6e-5, "
                             "encoder frozen, DropPath/Dropout2d active") if head == "train"
                            else "inference = fwd + softmax/argmax")),
            "per_gpu_batch": args.batch, "global_batch": args.batch * world, "num_classes": 5,
            "parallelism": f"dp{world}" + (" (DDP over RCCL + SyncBatchNorm)" if world > 1 else ""),
            "weights": "random init", "inputs_resident_in_hbm": True,
        },
    }
    if "train" in res and "infer" in res:
        out["inference_tiles_per_s"] = round(tiles / res["infer"], 3)
        out["inference_ms_per_step"] = round(1e3 * res["infer"] / args.steps, 3)
    # whole-model algorithmic flops: 1606.7 GF/tile train (frozen encoder), 726.7 fwd
    gf = {"dofa": {"train": 1606.7, "infer": 726.7}, "segformer": {"train": 3 * 121.0, "infer": 121.0},
          "unetpp": {"train": 3 * UNETPP_R18_FWD_GF, "infer": UNETPP_R18_FWD_GF}}[args.model]
    peak = PEAK_BF16_TFLOPS if use_bf16 else PEAK_F32_TFLOPS
    out["model_flops_utilisation"] = {
        k: round(gf[k] * 1e-3 * tiles / res[k] / world / peak, 4) for k in res}
    if timer is not None:
        summ = timer.summary()
        dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        name, s = dom
        achieved = s["flops"] / (s["ms"] * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
            "launches": s["launches"], "avg_launch_us": round(1e3 * s["ms"] / s["launches"], 2),
            "algorithmic_gflop_per_launch_avg": round(s["flops"] / s["launches"] / 1e9, 3),
            "share_of_step_time": round(s["ms"] * 1e-3 / res["train"], 4),
            "other_conv_gemm_variants": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                         for k, v in summ.items() if k != name},
        }
    if pcie is not None:
        out["pcie_inclusive"] = pcie
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    json_out.write(json.dumps(out) + "\n")
    json_out.flush()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
