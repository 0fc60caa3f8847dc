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

The stdout line is the SHORT form (< ~3 KB: contract keys, `roofline`, `cpu_baseline`, then one number per extra, the most
important ones last -- see compact_line); the LONG form with every table below goes to bench_details.json beside this file
(and to gpurun_out/bench_details.json when that directory exists).  `value` always covers exactly --steps steps; when those
take less than --min-seconds (3 s) the same steps are timed again over a longer run and reported under `sustained`.

Besides the contract fields the long form carries (N = 1 only, all measured in this same process):
  roofline      dominant MFMA kernel by HIP events on the launch stream; `traffic` = HBM bytes per launch from the
                committed rocprofv3 PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), see --help
  hbm_kernels   the HBM-bound kernels of the step at their DOFA shapes: algorithmic GB/s vs the 8 TB/s peak
  by_batch      the same train / inference step at per-GPU batch 2 / 4 / 8 (4 = the reference config's, dofa_config_RGB.yaml:85):
                headline numbers = the trainer's default path there (hipGraph replay), `eager` = the step launched from Python
  other_models  SegFormer-B2 (configs[2]) and UNet++/ResNet18 (configs[0]) steps at batch 32 (SIDE_BATCH: the side tables keep the batch of rounds 1-5)
  cpu_baseline  the CPU oracle on this box's cores: 2 warm-ups, median of 5
"""

from __future__ import annotations

import argparse
import math
import json
import os
import statistics
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
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak (6.3 TB/s measured achievable), same table
UNETPP_R18_FWD_GF = 128.04  # sum over convs of 2*K*C*R*S*Hout*Wout at 512x512 (counted on oracle/unetpp.py)
RGB_MEAN = [0.3992, 0.4283, 0.3998]   # configs/dofa_config_RGB.yaml:91-98
RGB_STD = [0.1672, 0.1800, 0.1584]
WAVELENGTHS = [0.665, 0.549, 0.481]   # configs/dofa_config_RGB.yaml:50
# whole-model algorithmic flops per tile: train (frozen encoder for DOFA, everything for the others), forward
MODEL_GF = {"dofa": {"train": 1606.7, "infer": 726.7}, "segformer": {"train": 3 * 121.0, "infer": 121.0},
            "unetpp": {"train": 3 * UNETPP_R18_FWD_GF, "infer": UNETPP_R18_FWD_GF},
            # configs[3]: +1.2 GF for the three extra bands of the dynamic patch embedding; configs[4]: SURVEY 8(d), per 1024^2 tile
            "dofa6": {"train": 1607.9, "infer": 727.9}, "dofa_large": {"train": 14400.0, "infer": 8826.6}}
MODEL_NAME = {"segformer": "SegFormer-B2 (MiT-B2 + MLP decoder)", "unetpp": "UNet++ (ResNet18 encoder)",
              "dofa": "DOFA-base + UperNet", "dofa6": "DOFA-base + UperNet, 6 bands", "dofa_large": "DOFA-large + UperNet, 10 bands, 1024x1024"}
# (bands, tile size, wavelengths in um) of the DOFA configurations: SURVEY 8(d) -- the reference gives the RGB list only
WAVELENGTHS6 = [0.665, 0.549, 0.481, 0.842, 1.610, 2.190]
WAVELENGTHS10 = [0.490, 0.560, 0.665, 0.705, 0.740, 0.783, 0.842, 0.865, 1.610, 2.190]
MODEL_INPUT = {"dofa6": (6, 512, WAVELENGTHS6), "dofa_large": (10, 1024, WAVELENGTHS10)}
SIDE_BATCH = 32      # per-GPU batch of the side tables (hbm_kernels, other_models) and of by_batch's extra entry: the headline batch of rounds 1-5
PMC_TRAFFIC_FILE = ROOT / "profiles" / "pmc_dominant_kernel_traffic.json"   # written by tools/pmc_bench_traffic.py


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=90, help="timed steps (default: ~6 s of training + ~3.3 s of inference at batch 64)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64,
                    help="per-GPU batch (weak scaling).  64 since round 6 (rounds 1-5: 32, still reported under by_batch['32']): "
                         "the step's small maps (18 x 18 / 36 x 36 pyramid levels) and layer tails fill the 256 CUs better -- same "
                         "box, same code: 901 -> 949 train, 1710 -> 1753 inference tiles/s; 96 and 128 add under 1 %")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mode", default="both", choices=["both", "train", "infer"])
    ap.add_argument("--model", default="dofa", choices=["dofa", "segformer", "unetpp", "dofa6", "dofa_large"],
                    help="dofa = DOFA-base+UperNet (headline, configs[1]); segformer = SegFormer-B2 (configs[2], all parameters trainable); "
                         "unetpp = UNet++/ResNet18 (configs[0], the reference's CPU smoke case); dofa6 = configs[3] (6 bands); "
                         "dofa_large = configs[4] (DOFA-large, 10 bands, 1024x1024 tiles: use --batch 8)")
    ap.add_argument("--no-input-stage", action="store_true", help="skip the PCIe-inclusive leg (host tiles through DeviceInputStage)")
    ap.add_argument("--with-input-stage", action="store_true",
                    help="also time the train step fed by host uint8 tiles through DeviceInputStage (PCIe-inclusive; "
                         "reported beside `value`, never as `value`)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="take the multi-GPU code path (RCCL group, SyncBatchNorm, DDP) even with one rank (self-test)")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="when --steps steps take less than this, ALSO time a longer run of the same step (reported under `sustained`; "
                         "`value` always covers exactly --steps steps); 0 = off")
    ap.add_argument("--details", default=None, help="where the long form goes (default: bench_details.json beside bench.py, and "
                                                     "gpurun_out/bench_details.json when that directory exists)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip hbm_kernels / by_batch / other_models (profiling runs)")
    return ap.parse_args()


def synthetic_batch(batch: int, device, seed: int, model: str = "dofa"):
    """uint8 U{0..255} tile -> /255 -> standardise (HIP kernel); mask U{0..4}.  Extra bands repeat the RGB
    statistics cyclically."""
    from geo_deep_learning.utils.tensors import normalize_standardize_u8
    bands, size, wv = MODEL_INPUT.get(model, (3, 512, WAVELENGTHS))
    g = torch.Generator(device="cpu").manual_seed(seed)
    u8 = torch.randint(0, 256, (batch, bands, size, size), generator=g, dtype=torch.uint8).to(device)
    mask = torch.randint(0, 5, (batch, 1, size, size), generator=g, dtype=torch.int64).to(device)
    mean = torch.tensor([RGB_MEAN[i % 3] for i in range(bands)], device=device)
    std = torch.tensor([RGB_STD[i % 3] for i in range(bands)], device=device)
    image = normalize_standardize_u8(u8, mean, std)
    return {"image": image, "mask": mask, "wavelengths": torch.tensor(wv)}


def timed(fn, steps: int, warmup: int, world: int, device) -> float:
    """W untimed + exactly K timed steps, barrier + synchronize on both sides, max over ranks."""
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)   # (cpu: the launch-path dry run)
    for _ in range(warmup):
        fn()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt


def build_task(model: str, device, dist_on: bool, local: int, capturable: bool = False):
    from gdlhip.nn import DiceLoss, FusedAdam
    opt = lambda params: FusedAdam(params, lr=6e-5, max_grad_norm=1.0, capturable=capturable)  # noqa: E731
    if model == "unetpp":
        from tasks_with_models.segmentation_unetplus import SegmentationUnetPlus
        task = SegmentationUnetPlus(encoder="resnet18", image_size=(512, 512), in_channels=3, num_classes=5,
                                    max_samples=6, loss=DiceLoss(mode="multiclass"), optimizer=opt)
    elif model == "segformer":
        from tasks_with_models.segmentation_segformer import SegmentationSegformer
        task = SegmentationSegformer(encoder="mit_b2", in_channels=3, num_classes=5, max_samples=6,
                                     loss=DiceLoss(mode="multiclass"), optimizer=opt)
    else:
        from tasks_with_models.segmentation_dofa import SegmentationDOFA
        size = MODEL_INPUT.get(model, (3, 512, None))[1]
        task = SegmentationDOFA(encoder="dofa_large" if model == "dofa_large" else "dofa_base", pretrained=False,
                                image_size=(size, size), num_classes=5, max_samples=6, loss=DiceLoss(mode="multiclass"),
                                freeze_layers=["encoder"], optimizer=opt)
    task.configure_model()
    task.to(device)
    if dist_on:
        # Lightning's `sync_batchnorm: true` + DDPStrategy(gradient_as_bucket_view=true)
        task.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(task.model)
        # (device_ids=None: with it set, DDP's forward moves the host-side `wavelengths` tensor to the GPU and the encoder reads it
        # back with a blocking copy every step -- round 5 finding, see gdlhip/trainer.py)
        ddp_kw = dict(gradient_as_bucket_view=True, find_unused_parameters=False)
        if capturable:
            # a later whole-step hipGraph capture needs wrapper construction, warm-up and capture on ONE side stream: main() makes
            # that stream (ddp.gdl_stream) the current one for everything it times
            from gdlhip.graphs import ddp_on_side_stream
            task.model = ddp_on_side_stream(task.model, **ddp_kw)
        else:
            task.model = torch.nn.parallel.DistributedDataParallel(task.model, **ddp_kw)
    (optimizer,), _ = task.configure_optimizers()
    return task, optimizer


def make_steps(task, optimizer, get_batch, use_bf16: bool):
    from gdlhip.markers import rng      # roctx ranges (GDL_ROCTX=1): forward groups in the model, loss / backward / optimizer here

    def train_step():
        task.train()
        optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16), rng("forward+loss"):
            loss = task.training_step(get_batch(), 0)
        with rng("backward"):
            loss.backward()
        with rng("optimizer"):
            optimizer.step()

    def infer_step():
        task.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
            task.validation_step(get_batch(), 0)
    return train_step, infer_step


class OpAccounting:
    """Roofline accounting of one step at the level of the C-ABI wrappers (gdlhip.ops): every top-level op call is bracketed
    by HIP events on the launch stream and charged its ALGORITHMIC bytes -- each tensor argument read once, each result
    written once, at its dtype; intermediates inside a fused op are not charged, weights and optimizer state are -- and,
    for the matrix ops, its algorithmic flops.  bound = max(bytes / 8 TB/s, flops / 2.5 PF/s); the fraction of a step is
    sum of the ops' bounds / measured step time.  An op called from inside another op of gdlhip.ops is charged to the outer
    call only (its tensors are the outer op's intermediates)."""

    OUT_KW = ("out", "dw", "din", "aux_out", "dq", "dk", "dv")
    SKIP = {"dt", "as_nhwc", "as_nchw", "image_f32", "split_qkv", "flash_ok", "resize_conv3x3_fwd_ok", "resize_conv3x3_fwd_bn_ok",
            "check", "KernelTimer", "resize_conv3x3_any_ok", "resize_conv3x3_bwd_gather_bn_ok", "bn_small_ok", "bn_small_fits",
            "dice_lowres_ok"}

    def __init__(self) -> None:
        self.records: list = []
        self._saved: dict = {}
        self._depth = 0

    @staticmethod
    def _nbytes(obj) -> int:
        if isinstance(obj, torch.Tensor):
            return obj.numel() * obj.element_size() if obj.is_cuda else 0
        if isinstance(obj, (list, tuple)):
            return sum(OpAccounting._nbytes(o) for o in obj)
        if isinstance(obj, dict):
            return sum(OpAccounting._nbytes(o) for o in obj.values())
        return 0

    @staticmethod
    def _flops(name: str, a: tuple, k: dict, ret) -> int:
        try:
            if name in ("conv_gemm", "linear"):
                x, w = a[0], a[1]
                out = ret[0] if isinstance(ret, tuple) else ret        # conv_gemm(want_stats=True) returns (out, partials, rows)
                m = (out.numel() // w.shape[0]) if isinstance(out, torch.Tensor) else 0
                return 2 * m * w.shape[0] * w.shape[1]
            if name == "conv_wgrad":
                x, dy = a[0], a[1]
                return 2 * (dy.numel() // dy.shape[-1]) * dy.shape[-1] * k.get("R", 1) * k.get("S", 1) * x.shape[-1]
            if name in ("attention", "attention_flash", "attention_unfused", "attention_flash_v1"):
                q, kk = a[0], a[1]
                return 4 * q.shape[0] * q.shape[1] * kk.shape[1] * q.shape[2]
            if name in ("attention_flash_bwd", "attention_bwd"):
                q, kk = a[0], a[1]
                return 10 * q.shape[0] * q.shape[1] * kk.shape[1] * q.shape[2]
            if name == "resize_conv3x3_bwd":      # two GEMMs over the low-resolution pixels (K = 9 N): data and weight gradient
                x_lo, dy, wd = a[0], a[1], a[2]
                px = x_lo.numel() // x_lo.shape[-1]
                g = 2 * px * x_lo.shape[-1] * 9 * dy.shape[-1]
                return g * ((wd is not None) + bool(k.get("want_dw", True)))
        except Exception:  # noqa: BLE001
            return 0
        return 0

    def __enter__(self):
        import types
        from gdlhip import ops
        for name, fn in list(vars(ops).items()):
            if isinstance(fn, types.FunctionType) and fn.__module__ == ops.__name__ and not name.startswith("_") and name not in self.SKIP:
                self._saved[name] = fn
                setattr(ops, name, self._wrap(name, fn))
        return self

    def _wrap(self, name, fn):
        def wrapper(*a, **k):
            if self._depth:                                  # an op called from inside another op (module globals are patched
                return fn(*a, **k)                           # too): charged to the outer call only
            self._depth += 1
            try:
                return charged(*a, **k)
            finally:
                self._depth -= 1

        def charged(*a, **k):
            reads = self._nbytes([v for v in a]) + self._nbytes({kk: v for kk, v in k.items() if kk not in self.OUT_KW})
            if name in ("multi_adam", "multi_sumsq"):        # chunk table rows [p, g, m, v, n]: 28 resp. 4 bytes per parameter
                reads = int(a[0][:, 4].sum().item()) * (28 if name == "multi_adam" else 4)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ret = fn(*a, **k)
            e1.record()
            writes = self._nbytes(ret) if ret is not None else self._nbytes({kk: v for kk, v in k.items() if kk in self.OUT_KW})
            self.records.append((name, reads + writes, self._flops(name, a, k, ret), e0, e1))
            return ret
        return wrapper

    def __exit__(self, *exc):
        from gdlhip import ops
        for name, fn in self._saved.items():
            setattr(ops, name, fn)
        return False

    def summary(self, step_ms: float, top: int = 3) -> dict:
        torch.cuda.synchronize()
        per: dict = {}
        for name, nbytes, flops, e0, e1 in self.records:
            d = per.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0, "flops": 0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nbytes
            d["flops"] += flops
        bound = lambda d: max(d["bytes"] / (PEAK_HBM_GBS * 1e9), d["flops"] / (PEAK_BF16_TFLOPS * 1e12)) * 1e3  # noqa: E731
        tot_b, tot_f = sum(d["bytes"] for d in per.values()), sum(d["flops"] for d in per.values())
        sum_bounds = sum(bound(d) for d in per.values())
        ranked = sorted(per.items(), key=lambda kv: -kv[1]["ms"])[:top]
        return {"algorithmic_gbytes_per_step": round(tot_b / 1e9, 3), "algorithmic_tflop_per_step": round(tot_f / 1e12, 3),
                "bound_ms": round(sum_bounds, 3), "bound_kind": "sum over ops of max(bytes / 8 TB/s, flops / 2.5 PF/s)",
                "whole_step_bound_ms": round(max(tot_b / (PEAK_HBM_GBS * 1e9), tot_f / (PEAK_BF16_TFLOPS * 1e12)) * 1e3, 3),
                "frac_of_bound": round(sum_bounds / step_ms, 4), "ops_ms_under_accounting": round(sum(d["ms"] for d in per.values()), 3),
                "top_ops": [{"op": n, "calls": d["calls"], "ms": round(d["ms"], 3), "gbytes": round(d["bytes"] / 1e9, 3),
                             "tflop": round(d["flops"] / 1e12, 3), "bound": "hbm" if d["bytes"] / (PEAK_HBM_GBS * 1e9) >= d["flops"] / (PEAK_BF16_TFLOPS * 1e12) else "mfma",
                             "frac_of_own_bound": round(bound(d) / max(d["ms"], 1e-9), 4)} for n, d in ranked]}


def side_measurement(model: str, batch_size: int, steps: int, warmup: int, device, use_bf16: bool, roofline: bool = False,
                     graphs: bool = False) -> dict:
    """Train + inference tiles/s of another model / batch size, measured like the headline (N = 1).  graphs: also with the
    whole step replayed from a hipGraph (gdlhip.graphs) -- at small batches the eager step is bound by the host issuing
    several hundred launches, not by the GPU."""
    task, optimizer = build_task(model, device, False, 0, capturable=graphs)
    batch = synthetic_batch(batch_size, device, 43, model)
    train_step, infer_step = make_steps(task, optimizer, lambda: batch, use_bf16)
    dt_t = timed(train_step, steps, warmup, 1, device)
    dt_i = timed(infer_step, steps, warmup, 1, device)
    peak = PEAK_BF16_TFLOPS if use_bf16 else PEAK_F32_TFLOPS
    n = batch_size * steps
    out = {"per_gpu_batch": batch_size, "tile": "x".join(str(v) for v in batch["image"].shape[1:]),
           "train_tiles_per_s": round(n / dt_t, 2), "inference_tiles_per_s": round(n / dt_i, 2),
           "train_ms_per_step": round(1e3 * dt_t / steps, 3), "inference_ms_per_step": round(1e3 * dt_i / steps, 3),
           "model_flops_utilisation": {"train": round(MODEL_GF[model]["train"] * 1e-3 * n / dt_t / peak, 4),
                                       "infer": round(MODEL_GF[model]["infer"] * 1e-3 * n / dt_i / peak, 4)}}
    if graphs:
        from gdlhip.graphs import GraphedEvalStep, GraphedTrainStep
        try:
            gt = GraphedTrainStep(task, optimizer, batch, autocast_dtype=torch.bfloat16 if use_bf16 else None)
            dt_g = timed(lambda: gt(), steps, warmup, 1, device)
            task.eval()
            ge = GraphedEvalStep(lambda b: task.validation_step(b, 0), batch, autocast_dtype=torch.bfloat16 if use_bf16 else None)
            dt_ge = timed(lambda: ge(), steps, warmup, 1, device)
            out["hipgraph"] = {"train_tiles_per_s": round(n / dt_g, 2), "train_ms_per_step": round(1e3 * dt_g / steps, 3),
                               "inference_tiles_per_s": round(n / dt_ge, 2), "inference_ms_per_step": round(1e3 * dt_ge / steps, 3),
                               "note": "forward + loss + backward + clip + Adam (resp. forward + argmax) captured once, replayed per step"}
            del gt, ge
            # the trainer's default at per-GPU batch <= 8 IS the captured step (MiniTrainer(graph_step="auto")): the entry's
            # headline numbers are that path's, the eager step (host-bound at these sizes: 312-450 tiles/s at batch 4
            # depending on the box's CPU, for the same 8.5 ms of GPU work) is reported beside it
            out["eager"] = {k: out[k] for k in ("train_tiles_per_s", "inference_tiles_per_s", "train_ms_per_step", "inference_ms_per_step")}
            out.update({k: v for k, v in out["hipgraph"].items() if k != "note"})
            out["default_path"] = "hipgraph replay (MiniTrainer(graph_step='auto') at per-GPU batch <= 8); `eager` = the same step launched from Python"
            out["model_flops_utilisation"] = {"train": round(MODEL_GF[model]["train"] * 1e-3 * n / dt_g / peak, 4),
                                              "infer": round(MODEL_GF[model]["infer"] * 1e-3 * n / dt_ge / peak, 4)}
        except Exception as exc:  # noqa: BLE001  (report, do not lose the eager numbers)
            out["hipgraph"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if roofline:
        # these models are HBM- / launch-bound, not MFMA-bound: a FLOP utilisation says little; one accounted step each
        for key, fn, ms in (("train", train_step, 1e3 * dt_t / steps), ("infer", infer_step, 1e3 * dt_i / steps)):
            with OpAccounting() as acct:
                fn()
            out.setdefault("roofline", {})[key] = acct.summary(ms)
    del task, optimizer, batch
    torch.cuda.empty_cache()
    return out


def hbm_kernels(device, b: int = 32) -> dict:
    """The HBM-bound kernels of one DOFA training step, each at its largest shape in the model (batch b), timed with
    HIP events on torch's current stream (the stream the kernels are launched on).  Bytes are ALGORITHMIC: every operand
    read once, every result written once at its stated dtype; fraction is of the 8 TB/s HBM3E peak."""
    from gdlhip import ops
    from gdlhip.nn import FusedAdam
    bf = torch.bfloat16
    res = {}

    def run(name, fn, nbytes, iters=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        gbs = nbytes / us / 1e3
        res[name] = {"us": round(us, 1), "algorithmic_mb": round(nbytes / 1e6, 1), "gb_per_s": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 3)}
    tok = torch.randn(b * 1297, 768, device=device)
    g, be = torch.ones(768, device=device), torch.zeros(768, device=device)
    run("layernorm_fwd f32->bf16 [B*1297,768]", lambda: ops.layernorm(tok, g, be, 1e-5, bf), tok.numel() * 6)
    y = torch.randn(b, 144, 144, 256, device=device).to(bf)
    n = y.numel()
    mean, var = ops.bn_stats(y)
    ga, bb = torch.ones(256, device=device), torch.zeros(256, device=device)
    run("bn_stats bf16 [B,144,144,256]", lambda: ops.bn_stats(y), n * 2)
    run("bn_apply+relu bf16 [B,144,144,256]", lambda: ops.bn_apply(y, mean, var, ga, bb, 1e-5, True), n * 4)
    dy = torch.randn_like(y)
    run("bn_bwd_reduce bf16 [B,144,144,256]", lambda: ops.bn_bwd_reduce(y, dy, mean, var, ga, bb, 1e-5, True), n * 4)
    sg, sb = ops.bn_bwd_reduce(y, dy, mean, var, ga, bb, 1e-5, True)
    run("bn_bwd_dx bf16 [B,144,144,256]", lambda: ops.bn_bwd_dx(y, dy, mean, var, ga, bb, 1e-5, True, sg, sb, n // 256), n * 6)
    x36 = torch.randn(b, 36, 36, 768, device=device).to(bf)
    up = torch.empty(b, 144, 144, 768, device=device, dtype=bf)
    run("bilinear_fwd bf16 36->144 x768", lambda: ops.bilinear(x36, (144, 144), out=up), (x36.numel() + up.numel()) * 2)
    run("bilinear_bwd bf16 144->36 x768", lambda: ops.bilinear_bwd(up, (36, 36)), (x36.numel() + up.numel()) * 2)
    low = torch.randn(b, 144, 144, 5, device=device)
    run("upsample_logits f32 144->512 x5", lambda: ops.upsample_logits(low, (512, 512)), (low.numel() + b * 5 * 512 * 512) * 4)
    logits = torch.randn(b, 5, 512, 512, device=device)
    tgt = torch.randint(0, 5, (b, 512, 512), device=device)
    run("dice_loss_fwd f32 [B,5,512,512]", lambda: ops.dice_loss_fwd(logits, tgt), logits.numel() * 4 + tgt.numel() * 8)
    _, sums = ops.dice_loss_fwd(logits, tgt)
    one = torch.ones((), device=device)
    run("dice_loss_bwd f32 [B,5,512,512]", lambda: ops.dice_loss_bwd(logits, tgt, sums, one), logits.numel() * 8 + tgt.numel() * 8)
    run("softmax_argmax f32 -> int64", lambda: ops.softmax_argmax(logits), logits.numel() * 4 + tgt.numel() * 8)
    u8 = torch.randint(0, 256, (b, 3, 512, 512), device=device, dtype=torch.uint8)
    m3, s3 = torch.tensor(RGB_MEAN, device=device), torch.tensor(RGB_STD, device=device)
    run("normalize_u8 -> f32 [B,3,512,512]", lambda: ops.normalize_u8(u8, m3, s3), u8.numel() * 5)
    p = torch.nn.Parameter(torch.randn(35_023_882, device=device))
    p.grad = torch.randn_like(p)
    opt = FusedAdam([p], lr=6e-5, max_grad_norm=1.0)
    run("fused Adam + clip, 35.0 M params", opt.step, p.numel() * (4 + 28), iters=5)   # sumsq pass reads g once more
    return res


def cpu_baseline(warm: int = 2, reps: int = 5):
    """The oracle (CPU restatement of the reference path, validated against reference goldens) timed on this box's
    host cores at batch 2, f32: `warm` untimed + median of `reps` timed steps for training and for inference."""
    import oracle
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    m = oracle.DOFASegmentationModel("dofa_base", (512, 512), num_classes=5, freeze_layers=["encoder"])
    b = oracle.synthetic_batch(2, 3, 512, 5, 42)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=6e-5)

    def train():
        opt.zero_grad(set_to_none=True)
        loss = oracle.model.training_loss(m(b["image"], b["wavelengths"]), b["mask"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()

    def infer():
        with torch.no_grad():
            oracle.model.predict_mask(m(b["image"], b["wavelengths"]))

    def med(fn):
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            fn()
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return statistics.median(ts)
    m.train()
    t_train = med(train)
    m.eval()
    t_inf = med(infer)
    return {"value": round(2 / t_train, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": f"train step (fwd+bwd+clip+Adam) at batch 2, f32: {warm} warm-ups, median of {reps} = {t_train:.2f} s; "
                      f"inference (forward+argmax) at batch 2: median {t_inf:.2f} s = {2 / t_inf:.3f} tiles/s",
            "inference_value": round(2 / t_inf, 4)}


def compact_line(out: dict, details: list) -> dict:
    """The ONE stdout line, kept under ~3 KB so that a driver that stores only a tail of stdout still holds every headline
    value; everything else (by_layer, by_k_depth, other_models' op tables, hbm_kernels, ddp.bucket_ready, ...) is the long form
    in bench_details.json.  Key order: the contract's keys first, then the extras with the most important ones LAST (a stored
    tail of 2000 characters then holds inference rate, both utilisations, the sustained run, step-level roofline fractions,
    small batches and the other BASELINE configs)."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")
    line = {k: out[k] for k in head if k in out}
    if "config" in line:
        cfg = dict(line["config"])
        cfg["workload"] = cfg["workload"][:150]
        line["config"] = cfg
    sr = out.get("step_roofline") or {}
    r = out.get("roofline")
    if r:
        line["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches",
                                             "avg_launch_us", "algorithmic_bytes_per_launch_avg", "share_of_step_time") if k in r}
        line["roofline"]["kernel"] = str(r.get("kernel", ""))[:48]
        for k in ("l2_hit_rate", "mfma_busy_share"):
            if k in (r.get("pmc") or {}):
                line["roofline"][k] = r["pmc"][k]
        for k, v in sr.items():      # (scalars: the whole step against the sum over its ops of max(bytes / 8 TB/s, flops / 2.5 PF/s))
            line["roofline"][f"step_frac_of_bound_{k}"] = v.get("frac_of_bound")
        top = (r.get("by_layer") or [])[:4]
        line["roofline"]["by_layer_tflops"] = {str(e["shape"])[:40]: e["tflops"] for e in top}
        line["roofline"]["other_variants_tflops"] = {str(k)[:28]: v["tflops"] for k, v in (r.get("other_conv_gemm_variants") or {}).items()}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {**{k: v for k, v in cb.items() if k != "sample"}, "sample": cb["sample"][:150]}
    line["details"] = details
    hk = out.get("hbm_kernels")
    if hk:
        line["hbm_kernels_gb_per_s"] = {k.split(" ")[0]: v["gb_per_s"] for k, v in hk.items()}
    d = out.get("ddp")
    if d:
        line["ddp"] = {k: d[k] for k in ("ranks", "rccl_version", "syncbn_messages_per_step", "step_ms_with_grad_sync", "step_ms_no_sync",
                                         "exposed_grad_comm_ms", "grad_allreduce_ms_alone", "graphed") if k in d}
    om = out.get("other_models")
    if om:
        line["other_models"] = {k: {"train": v.get("train_tiles_per_s"), "infer": v.get("inference_tiles_per_s"), "b": v.get("per_gpu_batch"),
                                    "frac_of_bound": {kk: vv.get("frac_of_bound") for kk, vv in (v.get("roofline") or {}).items()}}
                                for k, v in om.items()}
    bb = out.get("by_batch")
    if bb:
        line["by_batch"] = {k: {"train": v.get("train_tiles_per_s"), "infer": v.get("inference_tiles_per_s"),
                                "eager_train": (v.get("eager") or {}).get("train_tiles_per_s")} for k, v in bb.items()}
    if "pcie_inclusive" in out:
        line["pcie_inclusive"] = {"train_tiles_per_s": out["pcie_inclusive"]["train_tiles_per_s"]}
    if sr:
        line["step_roofline"] = {k: {"frac_of_bound": v.get("frac_of_bound"), "bound_ms": v.get("bound_ms"),
                                     "top_op": (v.get("top_ops") or [{}])[0].get("op")} for k, v in sr.items()}
    if "sustained" in out:
        line["sustained"] = {k: v for k, v in out["sustained"].items() if k != "note"}
    for k in ("executed_flops_utilisation", "model_flops_utilisation", "inference_ms_per_step", "inference_tiles_per_s"):
        if k in out:
            line[k] = out[k]
    return line


def self_launch(args) -> int:
    """`python bench.py --gpus N` started as ONE process (no WORLD_SIZE in the environment): re-execute this file under
    torch.distributed.run with N ranks on this node, the way the driver launches it, and hand its exit code back.  Rank 0's JSON
    line passes through to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting the ranks myself: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes fails without it on this driver
    return subprocess.call(cmd, env=env)


def dry_run(args, json_out, world: int, rank: int) -> None:
    """GDL_BENCH_DRY_RUN=1: the launch path of `--gpus N` without a GPU -- process group (gloo), barrier-bracketed timing with the
    max over ranks, one JSON line from rank 0 -- around a toy CPU step.  tests/test_distributed_cpu.py drives it; the line says
    `dry_run` and carries no throughput claim."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    torch.manual_seed(42 + rank)
    net = torch.nn.Linear(64, 64)
    model = DDP(net) if world > 1 else net
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    x = torch.randn(args.batch, 64)

    def step():
        opt.zero_grad(set_to_none=True)
        model(x).square().mean().backward()
        opt.step()
    dt = timed(step, args.steps, args.warmup, world, "cpu")
    ranks = dist.get_world_size() if world > 1 else 1
    if rank == 0:
        json_out.write(json.dumps({"metric": "dry run of the bench.py launch path (toy CPU step, no throughput claim)", "dry_run": True,
                                   "value": round(args.batch * world * args.steps / dt, 3), "unit": "toy samples/s", "n_gpus": world,
                                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
                                   "scaling": "weak", "ddp": {"backend": "gloo", "ranks": ranks}}) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # RCCL prints a version banner to the C-level stdout of every rank: keep a private handle on the real stdout for
    # the ONE JSON line and send everything else written to fd 1 to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): measuring {world}", file=sys.stderr)
    if os.environ.get("GDL_BENCH_DRY_RUN") == "1":
        dry_run(args, json_out, world, rank)
        return
    if torch.cuda.device_count() < max(world, local + 1):
        if rank == 0:
            print(f"bench.py: --gpus {world} needs {world} visible devices (one process per GPU), this node shows "
                  f"{torch.cuda.device_count()}", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_on = world > 1 or args.force_ddp
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")     # (whole-step capture under DDP: gdlhip/graphs.py)
        dist.init_process_group("nccl", device_id=device)

    from gdlhip import ops

    torch.manual_seed(42 + rank)  # train.py:67 seeds 42
    # at the reference's own per-GPU batch (4) the DDP step is launch-bound: the line then also times the step replayed from a
    # hipGraph that contains the RCCL collectives (MiniTrainer's default there); needs the device-side optimizer state
    ddp_graph = dist_on and args.batch <= 8
    task, optimizer = build_task(args.model, device, dist_on, local, capturable=ddp_graph)
    if ddp_graph:
        torch.cuda.set_stream(task.model.gdl_stream)      # (see build_task)
    batch = synthetic_batch(args.batch, device, 42 + rank, args.model)
    use_bf16 = args.dtype == "bf16"
    train_step, infer_step = make_steps(task, optimizer, lambda: batch, use_bf16)

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
    # `value` times EXACTLY --steps steps (the driver's contract).  A caller that passes a small K (the driver: 20 steps = 0.7 s)
    # gets a second, longer measurement of the same steps beside it: as many steps as make the timed region >= 3 s
    sustained = {}
    if args.min_seconds > 0 and not args.no_extras:      # (profiling / A-B runs pass --no-extras: exactly their --steps, nothing else)
        for key, fn in (("train", train_step), ("infer", infer_step)):
            if key in res and res[key] < args.min_seconds:
                k_long = int(math.ceil(args.min_seconds / (res[key] / args.steps)))
                sustained[key] = (k_long, timed(fn, k_long, 0, world, device))

    step_roofline = None
    if not args.no_extras and world == 1:
        # op-level accounting of the headline steps (see OpAccounting): what the whole step looks like against the HBM / MFMA
        # bounds of its ops, beside the dominant-kernel roofline below
        step_roofline = {}
        for key, fn in (("train", train_step), ("infer", infer_step)):
            if key in res:
                with OpAccounting() as acct:
                    fn()
                step_roofline[key] = acct.summary(1e3 * res[key] / args.steps, top=5)

    ddp_info = None
    if dist_on and "train" in res:
        from gdlhip import nn as gnn
        # the exchange step on its own: one all-reduce of the trainable gradients' bytes (what DDP's buckets move per step)
        nparam = sum(p.numel() for p in task.parameters() if p.requires_grad)
        buf = torch.zeros(nparam, device=device)
        dt_c = timed(lambda: dist.all_reduce(buf), 5, 2, world, device)
        del buf
        # SyncBatchNorm messages of one step, and what the gradient exchange costs INSIDE the step: the same step with DDP's
        # all-reduce switched off (no_sync: gradients stay local) against the normal one -- the difference is the part of
        # the communication that backward does not hide
        gnn.SYNC_MESSAGES[:] = [0, 0]
        train_step()
        sync_msgs = list(gnn.SYNC_MESSAGES)
        ddp_mod = task.model
        def step_no_sync():
            with ddp_mod.no_sync():
                train_step()
        k_ab = max(3, min(args.steps, 5))
        dt_ns = timed(step_no_sync, k_ab, 1, world, device)
        dt_s = timed(train_step, k_ab, 1, world, device)
        # when does each gradient bucket become ready?  A communication hook stamps an event on the autograd stream the moment
        # DDP hands it a full bucket (then runs the stock all-reduce); times are relative to the start of backward.  Buckets
        # that only become ready at the very end of backward are the part of the exchange nothing can hide.
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        stamps: list = []

        stamping = [True]      # (a comm hook cannot be unregistered: after the measurement it only forwards to the stock all-reduce)

        def stamp_hook(state, bucket):
            if stamping[0] and not torch.cuda.is_current_stream_capturing():
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                stamps.append((bucket.index(), bucket.buffer().numel() * bucket.buffer().element_size(), ev))
            return default_hooks.allreduce_hook(state, bucket)
        bucket_ready = None
        try:
            ddp_mod.register_comm_hook(None, stamp_hook)
            task.train()
            per_step = []
            for _ in range(3):
                stamps.clear()
                optimizer.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
                    loss = task.training_step(batch, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                loss.backward()
                e1.record()
                optimizer.step()
                torch.cuda.synchronize()
                per_step.append(([(i, nb, round(e0.elapsed_time(ev), 3)) for i, nb, ev in stamps], round(e0.elapsed_time(e1), 3)))
            last, bwd_ms = per_step[-1]
            bucket_ready = {"backward_ms": bwd_ms, "bucket_ready_ms": [t for _, _, t in sorted(last)],
                            "bucket_bytes": [nb for _, nb, _ in sorted(last)],
                            "note": "ms from the start of backward until DDP has the bucket's last gradient (all-reduce launch); "
                                    "bucket 0 holds the LAST layers of the model (first gradients of backward)"}
        except Exception as exc:  # noqa: BLE001  (private-ish torch API; never lose the line over a diagnostic)
            bucket_ready = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        stamping[0] = False
        try:
            log = ddp_mod._get_ddp_logging_data()
            sizes = [x for x in str(log.get("bucket_sizes", "")).replace(",", " ").split() if x]
            buckets = {"count": len(sizes) or None, "bucket_sizes": sizes[:16], "bucket_cap_bytes": log.get("bucket_cap_bytes")}
        except Exception:  # noqa: BLE001  (private torch API)
            buckets = {"count": None, "bucket_cap_bytes": None}
        ddp_info = {"backend": "nccl (RCCL)", "ranks": dist.get_world_size(),
                    "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                    "grad_bytes": nparam * 4, "grad_allreduce_ms_alone": round(1e3 * dt_c / 5, 3),
                    "syncbn_messages_per_step": {"forward": sync_msgs[0], "backward": sync_msgs[1]},
                    "step_ms_with_grad_sync": round(1e3 * dt_s / k_ab, 3), "step_ms_no_sync": round(1e3 * dt_ns / k_ab, 3),
                    "exposed_grad_comm_ms": round(1e3 * (dt_s - dt_ns) / k_ab, 3), "buckets": buckets, "bucket_ready": bucket_ready}

    if ddp_info is not None and ddp_graph:
        from gdlhip.graphs import GraphedTrainStep
        failure, gt = None, None
        try:
            gt = GraphedTrainStep(task, optimizer, batch, autocast_dtype=torch.bfloat16 if use_bf16 else None)
        except Exception as exc:  # noqa: BLE001  (report, keep the line)
            import traceback
            failure = f"{type(exc).__name__}: {exc}"[:300]
            print("bench.py: DDP step capture failed:\n" + traceback.format_exc(), file=sys.stderr)
        if world > 1:      # all ranks replay or none does
            flag = torch.tensor([0.0 if failure else 1.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() < 1.0:
                failure, gt = failure or "the capture failed on another rank", None
        if gt is not None:
            k_g = max(args.steps, 20)
            dt_g = timed(lambda: gt(), k_g, 3, world, device)
            ddp_info["graphed"] = {"train_tiles_per_s": round(args.batch * world * k_g / dt_g, 2), "ms_per_step": round(1e3 * dt_g / k_g, 3),
                                   "eager_ms_per_step": ddp_info["step_ms_with_grad_sync"],
                                   "note": "whole DDP step (forward, SyncBN messages, backward with bucket all-reduces, clip, Adam) "
                                           "replayed from ONE hipGraph per rank"}
            del gt
        else:
            ddp_info["graphed"] = {"error": failure}

    pcie = None
    if not args.no_input_stage and "train" in res:
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
        staged_train, _ = make_steps(task, optimizer, lambda: next(it), use_bf16)
        dt = timed(staged_train, args.steps, args.warmup, world, device)
        pcie = {"train_tiles_per_s": round(args.batch * world * args.steps / dt, 3),
                "h2d_bytes_per_tile": stage.bytes_h2d // (n_total * args.batch),
                "note": "host uint8 tiles + int64 masks (shipped as uint8) -> pinned ring -> copy stream (2 batches ahead) -> "
                        "normalise kernel -> step; `value` stays the HBM-resident rate"}

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    tiles = args.batch * world * args.steps
    head = "train" if "train" in res else "infer"
    model_name = MODEL_NAME[args.model]
    cfg_name = {"segformer": "configs[2]", "unetpp": "configs[0]", "dofa": "configs[1]", "dofa6": "configs[3]", "dofa_large": "configs[4]"}[args.model]
    bands_, size_, _ = MODEL_INPUT.get(args.model, (3, 512, None))
    out = {
        "metric": f"{size_}x{size_} tiles/s, {model_name}, {head} step",
        "value": round(tiles / res[head], 3),
        "unit": "tiles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * res[head] / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": (f"{model_name}, {bands_}-band {size_}x{size_} ({cfg_name}): "
                         + (("training step = fwd + Dice + bwd (every parameter) + clip 1.0 + Adam 6e-5, "
                             "DropPath/Dropout2d active" if not args.model.startswith("dofa") else
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
    gf = MODEL_GF[args.model]
    peak = PEAK_BF16_TFLOPS if use_bf16 else PEAK_F32_TFLOPS
    out["model_flops_utilisation"] = {
        k: round(gf[k] * 1e-3 * tiles / res[k] / world / peak, 4) for k in res}
    if timer is not None:
        summ = timer.summary()
        dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        name, s = dom
        achieved = s["flops"] / (s["ms"] * 1e-3) / 1e12
        traffic, traffic_note, pmc_extra = None, "no PMC summary committed", {}
        if PMC_TRAFFIC_FILE.is_file():
            pm = json.loads(PMC_TRAFFIC_FILE.read_text())
            sys.path.insert(0, str(ROOT / "tools"))
            from gemm_source_hash import gemm_source_hash
            if pm.get("gemm_source_hash") != gemm_source_hash():
                # the committed counters were taken on other kernel sources than the ones that just ran: report nothing
                pm, traffic_note = {"kernels": []}, ("stale: profiles/pmc_dominant_kernel_traffic.json was measured on other "
                                                      "conv_gemm sources (gemm_source_hash differs); rerun tools/pmc_bench_traffic.sh")
            for ent in pm.get("kernels", [pm]):          # one entry per implicit-GEMM kernel class (older files: one kernel)
                if ent.get("kernel_substring", "\0") in name:
                    traffic, traffic_note = ent["hbm_bytes_per_launch"], pm.get("note", ent.get("note", ""))
                    pmc_extra = {k: ent[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "l2_hit_rate",
                                                     "mfma_busy_share") if k in ent}
        out["roofline"] = {
            "bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_note,
            "pmc": pmc_extra,
            "launches": s["launches"], "avg_launch_us": round(1e3 * s["ms"] / s["launches"], 2),
            "algorithmic_gflop_per_launch_avg": round(s["flops"] / s["launches"] / 1e9, 3),
            # operands read once + result written once at their dtypes: compare with `traffic` (counted HBM bytes per launch)
            "algorithmic_bytes_per_launch_avg": int(s["bytes"] / s["launches"]),
            "share_of_step_time": round(s["ms"] * 1e-3 / res["train"], 4),
            # the same kernel class split by reduction depth K = R*S*C of its launches (TF/s per bucket)
            "by_k_depth": {b: {"launches": v["launches"], "ms": round(v["ms"], 3),
                               "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)}
                           for b, v in sorted(s.get("by_k", {}).items())},
            # ... and by layer shape (M = pixels or tokens, N = outputs, K = R*S*C), heaviest first: `achieved` is the average
            # over THIS mix -- a class that loses its deep-K members to an algebraic rewrite, or gains short-K ones because the
            # planner found this tile faster for them, moves the average without any layer getting slower
            "by_layer": [{"shape": t, "launches": v["launches"], "ms": round(v["ms"], 3),
                          "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)}
                         for t, v in sorted(s.get("by_shape", {}).items(), key=lambda kv: -kv[1]["ms"])[:14]],
            "other_conv_gemm_variants": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                             "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                         for k, v in summ.items() if k != name},
        }
    if step_roofline:
        out["step_roofline"] = step_roofline
        # two different utilisations, never to be confused: `model_flops_utilisation` prices the step at the REFERENCE's flops
        # (1606.7 / 726.7 GF per tile); this one at the flops the step actually EXECUTES after the algebraic
        # rewrites (low-resolution forms of conv3x3(resize(x)): 58 % of the reference's training flops are never computed)
        out["executed_flops_utilisation"] = {
            k: round(v["algorithmic_tflop_per_step"] / (1e3 * res[k] / args.steps) * 1e3 / peak, 4) for k, v in step_roofline.items()}
    if pcie is not None:
        out["pcie_inclusive"] = pcie
    if ddp_info is not None:
        out["ddp"] = ddp_info
    if world == 1 and not args.no_extras and args.model == "dofa" and use_bf16:
        del task, optimizer
        torch.cuda.empty_cache()
        side_batch = min(args.batch, SIDE_BATCH)      # the side tables stay at the batch of rounds 1-5 (comparable across rounds)
        out["hbm_kernels"] = hbm_kernels(device, side_batch)
        side_steps = min(max(args.steps, 10), 20)
        out["by_batch"] = {str(bsz): side_measurement("dofa", bsz, side_steps, args.warmup, device, True, graphs=True)
                           for bsz in (2, 4, 8)}      # 4 = the per-GPU batch of the reference's own config
        if args.batch != SIDE_BATCH:                  # the headline batch of rounds 1-5, eager like the headline
            out["by_batch"][str(SIDE_BATCH)] = side_measurement("dofa", SIDE_BATCH, side_steps, args.warmup, device, True)
        out["other_models"] = {m: side_measurement(m, side_batch, side_steps, args.warmup, device, True, roofline=True)
                               for m in ("segformer", "unetpp")}
        # BASELINE configs[3] / configs[4] on one GPU: the 6-band DOFA-base step and the DOFA-large 10-band 1024^2 step (per-GPU
        # batch 8 = 32 tiles of 512^2 worth of pixels; N = 5330 tokens: attention is a third of the forward's flops there)
        out["other_models"]["dofa_base_6band"] = side_measurement("dofa6", side_batch, side_steps, args.warmup, device, True, roofline=True)
        out["other_models"]["dofa_large_1024_10band"] = side_measurement("dofa_large", max(1, side_batch // 4), max(3, side_steps // 2),
                                                                          min(2, args.warmup), device, True, roofline=True)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    if sustained:
        out["sustained"] = {("train" if k == "train" else "inference") + "_tiles_per_s": round(args.batch * world * n / dt, 3)
                            for k, (n, dt) in sustained.items()}
        out["sustained"].update({"steps": {k: n for k, (n, _) in sustained.items()},
                                 "timed_region_s": {k: round(dt, 3) for k, (_, dt) in sustained.items()},
                                 "note": "`value` covers exactly --steps steps; these are the same steps timed over >= "
                                         f"{args.min_seconds:g} s (barrier + synchronize on both sides, max over ranks)"})
    details_paths = [Path(args.details)] if args.details else [ROOT / "bench_details.json"] + (
        [ROOT / "gpurun_out" / "bench_details.json"] if (ROOT / "gpurun_out").is_dir() else [])
    written = []
    for dp in details_paths:
        try:
            dp.write_text(json.dumps(out, indent=1) + "\n")
            written.append(str(dp.relative_to(ROOT)) if dp.is_relative_to(ROOT) else str(dp))
        except OSError as exc:
            print(f"bench.py: could not write {dp}: {exc}", file=sys.stderr)
    line = compact_line(out, written)
    json_out.write(json.dumps(line, separators=(",", ":")) + "\n")
    json_out.flush()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
