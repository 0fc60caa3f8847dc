"""Autograd-aware fused ops + precision policy + packed-weight cache.

Everything here is host logic around libgdlhip.so: forward AND backward arithmetic run in the
HIP kernels (``gdlhip.ops``); ``torch.autograd.Function`` only records the graph so that the
reference's ``loss.backward()`` / DDP hooks keep working unchanged.
"""

from __future__ import annotations

import contextlib
import os

import logging
import weakref
from typing import NamedTuple

import torch
import torch.distributed as dist
from torch import Tensor, nn
from torch.autograd import Function

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, ACT_RESID_RELU  # noqa: F401


_log = logging.getLogger(__name__)
_WARNED_FP16 = False
_WARNED_FALLBACK: set = set()


def warn_unfused(what: str, why: str) -> None:
    """One log line per (path, reason) when a grid-limit / shape guard sends a fused node down its unfused path."""
    if (what, why) not in _WARNED_FALLBACK:
        _WARNED_FALLBACK.add((what, why))
        _log.warning("gdlhip: %s runs UNFUSED (%s): correct, but several times slower", what, why)


# ------------------------------------------------------------------ precision policy
# A/B switch (tools): GDL_LAUNCH_FUSION=0 restores one launch per DropPath draw / BatchNorm counter / parameter cast
LAUNCH_FUSION = os.environ.get("GDL_LAUNCH_FUSION", "1") != "0"
_COUNTER_BATCH: list | None = None


@contextlib.contextmanager
def counter_batch():
    """Inside this context the ``num_batches_tracked += 1`` of every train-mode BatchNorm (nn.BatchNorm2d.forward) is collected
    and applied by ONE multi-tensor launch at exit instead of one launch per layer (21 in DOFA + UperNet, 62 in UNet++); the
    model forwards wrap themselves in it.  Outside of it ``bump`` increments at once."""
    global _COUNTER_BATCH
    if _COUNTER_BATCH is not None or not LAUNCH_FUSION:      # nested: the outermost context applies
        yield
        return
    _COUNTER_BATCH = []
    try:
        yield
    finally:
        pending, _COUNTER_BATCH = _COUNTER_BATCH, None
        if pending:
            torch._foreach_add_(pending, 1)


def bump(counter: Tensor | None) -> None:
    if counter is None:
        return
    if _COUNTER_BATCH is None:
        counter.add_(1)
    else:
        _COUNTER_BATCH.append(counter)


_KEEP_CACHE: dict = {}


def drop_path_scales(probs: list[float], batch: int, device) -> list[tuple[Tensor | None, Tensor | None]]:
    """The per-sample DropPath scales (mask / keep, timm ``drop_path`` with scale_by_keep) of a whole encoder pass, two draws per
    block (attention branch, MLP branch), from ONE Bernoulli launch and ONE division instead of two launches per draw (46 per
    DOFA-base step; at the reference's batch of 4 these sub-5-us launches are 2 % of the step).  ``probs[i]`` = block i's drop
    probability; blocks with probability 0 get (None, None).  Rows of one [2 n, B] tensor, dense."""
    live = [i for i, p_ in enumerate(probs) if p_ > 0.0]
    out: list[tuple[Tensor | None, Tensor | None]] = [(None, None)] * len(probs)
    if not live:
        return out
    if not LAUNCH_FUSION:
        for i in live:
            out[i] = tuple((torch.empty(batch, device=device, dtype=torch.float32).bernoulli_(1.0 - probs[i]) / (1.0 - probs[i]))
                           for _ in (0, 1))
        return out
    key = (str(device), tuple(probs[i] for i in live))
    keep = _KEEP_CACHE.get(key)
    if keep is None:
        keep = torch.tensor([1.0 - probs[i] for i in live for _ in (0, 1)], dtype=torch.float32, device=device).view(-1, 1)
        _KEEP_CACHE[key] = keep
    scales = torch.bernoulli(keep.expand(-1, batch)) / keep
    for j, i in enumerate(live):
        out[i] = (scales[2 * j], scales[2 * j + 1])
    return out


def compute_dtype() -> torch.dtype:
    """bf16 MFMA path under ``torch.autocast('cuda', bfloat16/float16)``, exact-f32 MFMA otherwise.

    The reference selects precision through Lightning's ``precision:`` flag, i.e. autocast
    (configs/dofa_config_RGB.yaml:12); the kernels accumulate in f32 either way.
    """
    if torch.is_autocast_enabled("cuda"):
        global _WARNED_FP16
        if not _WARNED_FP16 and torch.get_autocast_dtype("cuda") == torch.float16:
            # the reference's `precision: 16-mixed` (configs/dofa_config_RGB.yaml:12): there is no fp16 kernel set here
            _log.warning("gdlhip: fp16 autocast is computed in bf16 (f32 accumulation) by the HIP kernels; no loss scaling is "
                         "needed or applied")
            _WARNED_FP16 = True
        return torch.bfloat16
    return torch.float32


# ------------------------------------------------------------------ packed-weight cache
_RAW_WRITES: dict = {}   # id(param) -> number of raw-pointer updates (tensor._version does not see them)
_CACHE: dict = {}
_CACHE_BUILDS = [0]      # how often an operand was (re)built: whoever keeps lists of cache entries (FusedAdam.refresh_derived) rescans on a change


def mark_updated(p: Tensor) -> None:
    """Called for every tensor a kernel rewrites through a raw pointer (fused optimizer: parameters; single-GPU
    BatchNorm statistics kernel: running_mean / running_var) -- ``tensor._version`` does not see those writes."""
    if id(p) not in _RAW_WRITES:
        weakref.finalize(p, _RAW_WRITES.pop, id(p), None)
    _RAW_WRITES[id(p)] = _RAW_WRITES.get(id(p), 0) + 1


def cached(params: tuple, kind: str, builder):
    """Memoise ``builder()`` on the identity + version of ``params`` (tensors).  Frozen
    parameters never change version, so e.g. the DOFA dynamic patch-embed kernel of a frozen
    encoder is generated once per sensor, not once per step.  An entry dies with the tensors it was
    built from (weakref finalizers), so the cache never pins the device tensors of a dead model."""
    key = (kind, *[id(p) for p in params])
    ver = tuple((p._version, _RAW_WRITES.get(id(p), 0), p.data_ptr()) for p in params)
    hit = _CACHE.get(key)
    # ids (and allocator addresses) are recycled once a model is garbage collected: an entry only counts when it
    # was built from these very tensor objects
    if hit is not None and hit[0] == ver and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    val = builder()
    _CACHE_BUILDS[0] += 1
    if hit is None:
        for p in params:
            weakref.finalize(p, _CACHE.pop, key, None)
    _CACHE[key] = (ver, val, tuple(weakref.ref(p) for p in params))
    return val


def conv_weight_matrix(weight: Tensor) -> Tensor:
    """[N,C,R,S] conv parameter -> f32 [N, R*S*C] view (free when stored channels_last)."""
    n = weight.shape[0]
    w = weight.detach().permute(0, 2, 3, 1)
    if not w.is_contiguous():
        w = w.contiguous()  # parameter not stored channels_last: one repack (cached by caller)
    return w.reshape(n, -1)


def gemm_weight(weight: Tensor, cd: torch.dtype) -> Tensor:
    """GEMM operand ([N, K] K-contiguous, compute dtype) of a Linear / conv parameter."""
    def build():
        w = weight.detach() if weight.dim() == 2 else conv_weight_matrix(weight)
        w = w if w.is_contiguous() else w.contiguous()
        return w if cd == torch.float32 else ops.cast(w, cd)
    return cached((weight,), f"gemm:{cd}", build)


def gemm_weight_shadow(weight: Tensor):
    """(cache key, bf16 tensor) of ``gemm_weight(weight, bf16)`` when that operand exists and is a plain cast of the parameter's
    dense storage (same element order) -- then the fused optimizer rewrites it together with the parameter
    (gdl_multi_adam's `shadow` column) and ``refresh_shadow`` revalidates the cache entry; None otherwise."""
    key = (f"gemm:{torch.bfloat16}", id(weight))
    hit = _CACHE.get(key)
    if hit is None or hit[2][0]() is not weight:
        return None
    val = hit[1]
    if val.dtype != torch.bfloat16 or val.numel() != weight.numel() or not val.is_contiguous() or weight.dtype != torch.float32:
        return None
    if weight.dim() == 2:
        same_order = weight.is_contiguous()
    else:
        same_order = weight.dim() == 4 and weight.permute(0, 2, 3, 1).is_contiguous()
    return (key, val) if same_order else None


def refresh_shadow(key, val: Tensor, weight: Tensor) -> None:
    """After the optimizer rewrote ``weight`` AND its bf16 shadow ``val``: the cache entry is current again."""
    hit = _CACHE.get(key)
    if hit is not None and hit[1] is val and hit[2][0]() is weight:
        _CACHE[key] = ((((weight._version, _RAW_WRITES.get(id(weight), 0), weight.data_ptr())),), val, hit[2])


# bf16 operands DERIVED from a conv parameter in another element order (channel slice, tap-major, data gradient): the fused
# optimizer rebuilds all of them in ONE launch behind its update (gdl_multi_repack) and revalidates their cache entries, like the
# plain bf16 copies above -- id(weight) -> {cache key: (mode, c0, c1)}
_DERIVED: dict = {}
REPACK_SLICE, REPACK_TAPS, REPACK_DGRAD = 0, 1, 2
REPACK_FUSION = os.environ.get("GDL_REPACK_FUSION", "1") != "0"   # A/B switch: 0 = derived operands are rebuilt on their next use


def _register_derived(weight: Tensor, kind: str, cd: torch.dtype, mode: int, c0: int, c1: int) -> None:
    if cd != torch.bfloat16 or not LAUNCH_FUSION or not REPACK_FUSION or weight.dtype != torch.float32 or not weight.requires_grad:
        return
    if not (weight.dim() == 4 and weight.permute(0, 2, 3, 1).is_contiguous()):
        return                                            # (the kernel reads the parameter's dense [N][T][C] storage)
    ent = _DERIVED.get(id(weight))
    if ent is None:
        ent = _DERIVED[id(weight)] = {}
        weakref.finalize(weight, _DERIVED.pop, id(weight), None)
    ent[(kind, id(weight))] = (mode, c0, c1)


def derived_operands(weight: Tensor) -> list:
    """[(cache key, bf16 tensor, mode, c0, c1)] of the derived operands of ``weight`` that exist in the cache right now."""
    out = []
    for key, (mode, c0, c1) in _DERIVED.get(id(weight), {}).items():
        hit = _CACHE.get(key)
        if hit is None or hit[2][0]() is not weight:
            continue
        val = hit[1]
        if isinstance(val, Tensor) and val.dtype == torch.bfloat16 and val.is_contiguous():
            out.append((key, val, mode, c0, c1))
    return out


def adopt_operand(key, val: Tensor, weight: Tensor) -> None:
    """``val`` was just rewritten from the CURRENT ``weight`` by a kernel that holds its address (a hipGraph replay of the
    optimizer's update): make it THE cache entry of ``key`` again, whatever an eager step put there in between.  Without this
    an eager step between two replays builds new operand tensors, the eager update rewrites those, and the next replay's forward
    reads operands that miss one update."""
    hit = _CACHE.get(key)
    if hit is None or hit[1] is not val:
        _CACHE_BUILDS[0] += 1
    if hit is not None and hit[2][0]() is weight:
        refs = hit[2]
    else:
        refs = (weakref.ref(weight),)
        if hit is None:
            weakref.finalize(weight, _CACHE.pop, key, None)
    _CACHE[key] = (((weight._version, _RAW_WRITES.get(id(weight), 0), weight.data_ptr()),), val, refs)


def dgrad_weight(weight: Tensor, cd: torch.dtype) -> Tensor:
    """Flipped / transposed operand for the data gradient of a conv parameter."""
    def build():
        n, c, r, s = weight.shape
        return ops.pack_dgrad(conv_weight_matrix(weight), n, r * s, c, cd)
    _register_derived(weight, f"dgrad:{cd}", cd, REPACK_DGRAD, 0, weight.shape[1])
    return cached((weight,), f"dgrad:{cd}", build)


class _ToCompute(Function):
    @staticmethod
    def forward(ctx, x, cd):
        ctx.in_dtype = x.dtype
        return ops.copy_cast(x, out_dtype=cd)

    @staticmethod
    def backward(ctx, g):
        return ops.copy_cast(g, out_dtype=ctx.in_dtype), None


def to_compute(x: Tensor, cd: torch.dtype) -> Tensor:
    """NHWC tensor in the compute dtype (strided copy+cast through the identity resize)."""
    if x.dtype == cd:
        return x
    if x.requires_grad and torch.is_grad_enabled():
        return _ToCompute.apply(x, cd)
    return ops.copy_cast(x, out_dtype=cd)


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


# ------------------------------------------------------------------ SyncBatchNorm exchange
FUSE_SYNC_MESSAGE = os.environ.get("GDL_SYNC_MESSAGE_KERNELS", "1") != "0"   # A/B switch: 0 = the message is built by torch expressions


def sync_batch_stats(mean: Tensor, var: Tensor, group=None, count: int | None = None):
    """Merge per-rank batch statistics into global ones (nn.SyncBatchNorm forward): count-weighted, so a
    ragged last batch (ranks with different pixel counts) merges exactly like torch's all_gather of
    [mean, invstd, count] does -- in ONE small all-reduce of 2C + 1 floats: sum_r n_r * [mean_r, E_r[x^2], 1].
    Returns (global mean, global biased variance, global count); the count stays a 0-dim DEVICE tensor (no host
    read-back: 21 layers per step would mean 21 stream syncs)."""
    n = float(count if count is not None else 1)
    c = mean.numel()
    if mean.is_cuda and FUSE_SYNC_MESSAGE:
        # one pack kernel, the all-reduce, one unpack kernel (the torch expressions below are ~12 one-line launches per layer)
        packed = torch.empty(2 * c + 1, device=mean.device, dtype=torch.float32)
        ops.syncbn_pack(mean, var, n, packed)
        dist.all_reduce(packed, group=group)
        SYNC_MESSAGES[0] += 1
        gmean, gvar = ops.syncbn_unpack(packed, c)
        return gmean, gvar, packed[2 * c]
    packed = torch.cat([mean * n, (var + mean * mean) * n, mean.new_full((1,), n)])
    dist.all_reduce(packed, group=group)
    SYNC_MESSAGES[0] += 1
    total = packed[2 * c]
    gmean = (packed[:c] / total).contiguous()
    gvar = (packed[c:2 * c] / total - gmean * gmean).clamp_min_(0).contiguous()
    return gmean, gvar, total


def update_running_stats(running_mean, running_var, mean, var, momentum: float, count) -> None:
    """nn.BatchNorm2d running-stat update with the UNBIASED variance; ``count`` int or 0-dim tensor."""
    running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
    if isinstance(count, Tensor):
        running_var.mul_(1 - momentum).add_(var * (momentum * count / (count - 1).clamp_min(1)))
    else:
        running_var.mul_(1 - momentum).add_(var, alpha=momentum * count / max(count - 1, 1))


def sync_sum_pair(a: Tensor, b: Tensor, group=None) -> tuple[Tensor, Tensor]:
    """all-reduce(SUM) of the two BN-backward reductions in one message (SyncBatchNorm backward)."""
    packed = torch.stack([a, b])
    dist.all_reduce(packed, group=group)
    SYNC_MESSAGES[1] += 1
    return packed[0].contiguous(), packed[1].contiguous()


FUSE_UP4 = True   # A/B switch: False = materialise the x4 upsample and run the ordinary 3x3 kernel
FUSE_TAPSUM = True   # A/B switch: False = round-2 forward of resized-input convolutions (sub-pixel phases / concat buffer)
FUSE_TAPSUM_STATS = True   # A/B switch: False = separate BatchNorm statistics pass over the gather-sum's output


def tap_weight(weight: Tensor, cd: torch.dtype, c0: int = 0, c1: int | None = None) -> Tensor:
    """[9 N, C'] operand of the low-resolution forward of conv3x3(resize(x)) (ops.resize_conv3x3_fwd_sum): row t * N + n
    is filter tap t = 3 r + s of output channel n over the input-channel slice c0:c1 -- the nine tap products
    z = [W_0 x, ..., W_8 x] are then ONE 1x1 convolution with 9 N output channels."""
    def build():
        n, c = weight.shape[0], weight.shape[1]
        w = conv_weight_matrix(weight).view(n, 9, c)[:, :, c0:c1].permute(1, 0, 2).reshape(9 * n, -1).contiguous()
        return w if cd == torch.float32 else ops.cast(w, cd)
    if tuple(weight.shape[2:]) == (3, 3):
        _register_derived(weight, f"tap:{cd}:{c0}:{c1}", cd, REPACK_TAPS, c0, weight.shape[1] if c1 is None else c1)
    return cached((weight,), f"tap:{cd}:{c0}:{c1}", build)


def slice_weight(weight: Tensor, cd: torch.dtype, c0: int, c1: int) -> Tensor:
    """[N, 9 * (c1 - c0)] GEMM operand of a 3x3 conv parameter restricted to the input channels c0:c1 (one level of a concat)."""
    def build():
        n, c = weight.shape[0], weight.shape[1]
        w = conv_weight_matrix(weight).view(n, 9, c)[:, :, c0:c1].reshape(n, -1).contiguous()
        return w if cd == torch.float32 else ops.cast(w, cd)
    if tuple(weight.shape[2:]) == (3, 3):
        _register_derived(weight, f"slice:{cd}:{c0}:{c1}", cd, REPACK_SLICE, c0, c1)
    return cached((weight,), f"slice:{cd}:{c0}:{c1}", build)


# ------------------------------------------------------------------ conv -> BN -> ReLU
# messages of the SyncBatchNorm exchange since the last reset: [forward all-reduces, backward all-reduces]
SYNC_MESSAGES = [0, 0]


def _cba_conv(x, weight, cb, pad, up4, sync_group, running_mean, running_var, momentum):
    """conv(+bias) of a training ConvModule on NHWC x (the resize of an `up4` member fused in) -> (y, (mean, var) | None):
    the batch statistics come with y when the producing kernel emits them (gather-sum of the low-resolution forward)."""
    cd = x.dtype
    n, c, r, s = weight.shape
    if up4 and FUSE_TAPSUM and ops.resize_conv3x3_fwd_ok((x.shape[1], x.shape[2]), (up4 * x.shape[1], up4 * x.shape[2]), x.shape[0]):
        # conv3x3(resize(x)) = sum_t shift_t(resize(W_t x)): nine tap products as ONE 1x1 convolution over the
        # LOW-resolution pixels (1 / up4^2 of the MACs), then one gather-sum pass writes the full-resolution output
        z, size = ops.conv_gemm(x, tap_weight(weight, cd)), (up4 * x.shape[1], up4 * x.shape[2])
        if FUSE_TAPSUM_STATS and ops.resize_conv3x3_fwd_bn_ok(cd, n, cb):
            # ... and the batch statistics of the output come out of the same pass (per-block partial sums)
            own = (_world(sync_group) if sync_group is not False else 1) == 1
            y, mean, var = ops.resize_conv3x3_fwd_sum_bn([z], size, addvec=cb, running_mean=running_mean if own else None,
                                                         running_var=running_var if own else None, momentum=momentum)
            return y, (mean, var)
        return ops.resize_conv3x3_fwd_sum([z], size, addvec=cb), None
    if up4 == 4:   # conv3x3(bilinear_x4(x)) without the upsampled intermediate (ops.up4_conv3x3)
        return ops.up4_conv3x3(x, subpix4_weight(weight, cd), bias=cb), None
    if up4:        # other resize factors: the upsampled map is a temporary of the forward only (backward works on x)
        return ops.conv_gemm(ops.bilinear(x, (up4 * x.shape[1], up4 * x.shape[2])), gemm_weight(weight, cd), R=r, S=s, pad=pad,
                             bias=cb), None
    small = (sync_group is False or _world(sync_group) == 1) and x.dim() == 4 and ops.BN_SMALL_MAX_PIXELS > 0 and \
        x.shape[0] * ((x.shape[1] + 2 * pad - r) + 1) * ((x.shape[2] + 2 * pad - s) + 1) <= ops.BN_SMALL_MAX_PIXELS
    if FUSE_CONV_STATS and cd == torch.bfloat16 and not small:     # (small maps: the whole BatchNorm is ONE launch, ops.bn_small_fwd)
        # the batch statistics as a side output of the convolution's epilogue (per-wave partial sums of the bf16 outputs)
        # wherever the launch is made of whole tiles on the coalesced-epilogue kernels: no statistics pass over y
        y, partials, rows = ops.conv_gemm(x, gemm_weight(weight, cd), R=r, S=s, pad=pad, bias=cb, want_stats=True)
        if not rows:
            return y, None
        own = (_world(sync_group) if sync_group is not False else 1) == 1
        return y, ops.bn_stats_finalize(partials, rows, n, y.numel() // n, running_mean if own else None,
                                        running_var if own else None, momentum)
    return ops.conv_gemm(x, gemm_weight(weight, cd), R=r, S=s, pad=pad, bias=cb), None


FUSE_CONV_STATS = os.environ.get("GDL_CONV_STATS", "1") != "0"   # A/B switch: False / GDL_CONV_STATS=0 = a separate statistics pass over every convolution output (round 3)


# A/B switch, OFF by default: measured SLOWER than the two launches it replaces (round 5, same-box A/B, profiles/r05a_*: 881-883
# vs 890-892 train tiles/s).  The gather's windows overlap 3.4x (10 x 22 output pixels staged per 4 x 16 footprint): the LDS-DMA
# re-reads them from L2 for free, the register-staged BN form recomputes ~10 VALU operations per element 3.4 times with two
# waves per SIMD and no DMA overlap.  Kept with its parity test as the measured reference for that design.
FUSE_BN_BWD_GATHER = os.environ.get("GDL_FUSE_BN_BWD_GATHER", "0") == "1"


def _bn_bwd_and_grads(x, weight, y, gout, mean, var, g, b, eps, relu, sg, sb, p_local, pad, up4, need_dx, need_dw):
    """BatchNorm(+ReLU) backward of a training ConvModule followed by its convolution's gradients.  For the resized 3x3
    members (up4) the BN backward is formed inside the gather kernel that consumes it (round 5: one HBM pass instead of
    bn_bwd_dx's three + the gather's read); everything else writes dy in place of y and hands it to _cba_grads."""
    if (up4 and FUSE_BN_BWD_GATHER and (need_dx or need_dw) and y.dim() == 4 and gout.is_contiguous()
            and ops.resize_conv3x3_bwd_gather_bn_ok(gout, (x.shape[1], x.shape[2]))):
        maps = ops.resize_conv3x3_bwd_gather_bn(gout, y, (x.shape[1], x.shape[2]), mean, var, g, b, eps, relu, sg, sb, p_local)
        return _cba_grads(x, weight, None, pad, up4, need_dx, need_dw, maps=maps)
    dy = ops.bn_bwd_dx(y, gout, mean, var, g, b, eps, relu, sg, sb, p_local, out=y)
    return _cba_grads(x, weight, dy, pad, up4, need_dx, need_dw)


def _cba_grads(x, weight, dy, pad, up4, need_dx, need_dw, maps=None):
    """(dx, dw) of the conv of a training ConvModule from the gradient dy of its output (BatchNorm backward already applied)."""
    n, c, r, s = weight.shape
    dx = dw = None
    if up4:
        # both gradients as GEMMs over the LOW-resolution pixels: the resize and the tap shifts act on pixels, the
        # weights on channels, so dx = sum_t W_t^T G_t and dW_t = sum_q G_t[q] (x) x[q] with G_t = resize^T shift_t^T dy
        # (ops.resize_conv3x3_bwd): 1/16 of the MACs of the full-resolution data gradient + phase weight gradients
        dx, dw = ops.resize_conv3x3_bwd(x, dy, dgrad_weight(weight, x.dtype) if need_dx else None, want_dw=need_dw, g=maps)
    else:
        if need_dx:
            dx = ops.conv_gemm(dy, dgrad_weight(weight, x.dtype), R=r, S=s, pad=r - 1 - pad)
        if need_dw:
            dw = ops.conv_wgrad(x, dy, R=r, S=s, pad=pad)
    if dw is not None:
        # same strides as the channels-last parameter (for 1x1 kernels torch keeps (C,1,1,1))
        dw = dw.view(n, c, 1, 1) if r == 1 and s == 1 else dw.view(n, r, s, c).permute(0, 3, 1, 2)
    return dx, dw


class _ConvBNActTrain(Function):
    """Training-mode ConvModule: conv(+bias) -> BatchNorm(batch stats) -> ReLU.

    Reference: models/utils.py:10-52, multilevel_neck.py:28-67 with nn.BatchNorm2d /
    nn.SyncBatchNorm semantics.
    """

    @staticmethod
    def forward(ctx, x, weight, conv_bias, gamma, beta, running_mean, running_var, momentum, eps,
                pad, relu, sync_group, up4=False):
        n = weight.shape[0]
        cb = None if conv_bias is None else conv_bias.detach()
        y, stats_done = _cba_conv(x, weight, cb, pad, up4, sync_group, running_mean, running_var, momentum)
        world = _world(sync_group) if sync_group is not False else 1
        p_local, p_share = y.numel() // n, None
        if world > 1:
            mean, var = stats_done if stats_done is not None else ops.bn_stats(y)
            mean, var, total = sync_batch_stats(mean, var, sync_group or None, count=p_local)
            if running_mean is not None:
                update_running_stats(running_mean, running_var, mean, var, momentum, total)
            p_share = p_local / total          # this rank's share of the global pixel count (device scalar)
        else:
            out = None
            if stats_done is None and ops.bn_small_ok(y):
                # a small map (the reference's own per-GPU batch of 4 makes most decoder layers small): statistics, running
                # estimates and the normalised output in ONE launch
                out, mean, var = ops.bn_small_fwd(y, gamma.detach(), beta.detach(), eps, relu, running_mean, running_var, momentum)
            else:
                mean, var = stats_done if stats_done is not None else ops.bn_stats(y, running_mean, running_var, momentum)
            if running_mean is not None:     # written through raw pointers: invalidate the eval-mode fold cache
                mark_updated(running_mean)
                mark_updated(running_var)
            if out is not None:
                ctx.save_for_backward(x, weight, y, mean, var, gamma, beta)
                ctx.cfg = (pad, relu, eps, conv_bias is not None, sync_group, world, p_local, p_share, up4)
                return out
        out = ops.bn_apply(y, mean, var, gamma.detach(), beta.detach(), eps, relu)
        ctx.save_for_backward(x, weight, y, mean, var, gamma, beta)
        ctx.cfg = (pad, relu, eps, conv_bias is not None, sync_group, world, p_local, p_share, up4)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, y, mean, var, gamma, beta = ctx.saved_tensors
        pad, relu, eps, has_bias, sync_group, world, p_local, p_share, up4 = ctx.cfg
        n = weight.shape[0]
        if gout.dtype != y.dtype:
            gout = to_compute(gout, y.dtype)
        g, b = gamma.detach(), beta.detach()
        if world == 1 and ops.bn_small_ok(y) and not (up4 and FUSE_BN_BWD_GATHER):
            # small map: sums, parameter gradients and dy (written over y) in ONE launch
            dy, dgamma, dbeta = ops.bn_small_bwd(y, gout, mean, var, g, b, eps, relu, out=y)
            dbias = torch.zeros(n, device=x.device, dtype=torch.float32) if has_bias and ctx.needs_input_grad[2] else None
            dx, dw = _cba_grads(x, weight, dy, pad, up4, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
            return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None
        dgamma, dbeta = ops.bn_bwd_reduce(y, gout, mean, var, g, b, eps, relu)
        sg, sb = dgamma, dbeta
        if world > 1:
            # the kernel divides the two sums by p_local: scaling them by p_local / P_global makes that the global mean
            sg, sb = sync_sum_pair(dgamma, dbeta, sync_group or None)
            sg, sb = sg * p_share, sb * p_share
        dbias = None
        if has_bias and ctx.needs_input_grad[2]:
            # a bias feeding train-mode BN has an analytically zero gradient
            dbias = torch.zeros(n, device=x.device, dtype=torch.float32)
        dx, dw = _bn_bwd_and_grads(x, weight, y, gout, mean, var, g, b, eps, relu, sg, sb, p_local, pad, up4,
                                   ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None


class _ConvBNActGroupTrain(Function):
    """K independent training ConvModules (conv -> SyncBatchNorm(batch stats) -> ReLU on K different inputs) as ONE autograd
    node, so that their statistics cross the ranks in ONE all-reduce forward (all members' [count * mean, count * E[x^2],
    count]) and ONE backward (all members' [sum dy, sum dy * xhat]) instead of K each: the four lateral / four 3x3
    convolutions of MultiLevelNeck (multilevel_neck.py:141-160), UperNet's laterals + PPM branches + auxiliary head
    convolution (upernet.py:103-127) and its three fpn_convs (:137-141) are siblings -- 21 + 21 latency-bound messages per
    step become 6 + 6.  Per member the arithmetic is _ConvBNActTrain's; autograd runs this node's backward once all K
    output gradients exist.  flat = K x (x, weight, conv_bias, gamma, beta, running_mean, running_var)."""

    @staticmethod
    def forward(ctx, metas, sync_group, *flat):
        k = len(metas)
        mem = [flat[7 * i:7 * i + 7] for i in range(k)]
        ys, locals_, counts = [], [], []
        for (x, weight, conv_bias, gamma, beta, rm, rv), (momentum, eps, pad, relu, up4) in zip(mem, metas):
            cb = None if conv_bias is None else conv_bias.detach()
            y, stats_done = _cba_conv(x, weight, cb, pad, up4, sync_group, rm, rv, momentum)
            ys.append(y)
            locals_.append(stats_done if stats_done is not None else ops.bn_stats(y))
            counts.append(y.numel() // weight.shape[0])
        # one message: per member [n * mean, n * E[x^2], n]  (count-weighted merge, see sync_batch_stats)
        fused = FUSE_SYNC_MESSAGE and ys[0].is_cuda
        if fused:      # every member packs its share straight into its slice of the message: no torch arithmetic, no cat
            packed = torch.empty(sum(2 * w.shape[0] + 1 for (_, w, *_r) in mem), device=ys[0].device, dtype=torch.float32)
            off = 0
            for (mean, var), n in zip(locals_, counts):
                ops.syncbn_pack(mean, var, float(n), packed[off:off + 2 * mean.numel() + 1])
                off += 2 * mean.numel() + 1
        else:
            packed = torch.cat([t for (mean, var), n in zip(locals_, counts)
                                for t in (mean * float(n), (var + mean * mean) * float(n), mean.new_full((1,), float(n)))])
        dist.all_reduce(packed, group=sync_group or None)
        SYNC_MESSAGES[0] += 1
        outs, saved, cfg, off = [], [], [], 0
        for (x, weight, conv_bias, gamma, beta, rm, rv), (momentum, eps, pad, relu, up4), y, n in zip(mem, metas, ys, counts):
            c = weight.shape[0]
            total = packed[off + 2 * c]
            if fused:  # global statistics + the running-estimate update in one kernel
                gmean, gvar = ops.syncbn_unpack(packed[off:off + 2 * c + 1], c, rm, rv, momentum)
                if rm is not None:
                    mark_updated(rm)
                    mark_updated(rv)
            else:
                gmean = (packed[off:off + c] / total).contiguous()
                gvar = (packed[off + c:off + 2 * c] / total - gmean * gmean).clamp_min_(0).contiguous()
                if rm is not None:
                    update_running_stats(rm, rv, gmean, gvar, momentum, total)
            off += 2 * c + 1
            outs.append(ops.bn_apply(y, gmean, gvar, gamma.detach(), beta.detach(), eps, relu))
            saved += [x, weight, y, gmean, gvar, gamma, beta]
            cfg.append((pad, relu, eps, conv_bias is not None, n, (total if fused else n / total), up4))
        ctx.fused_message = fused
        ctx.save_for_backward(*saved)
        ctx.cfg, ctx.sync_group = cfg, sync_group
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        k = len(ctx.cfg)
        mem = [ctx.saved_tensors[7 * i:7 * i + 7] for i in range(k)]
        sums = []
        gs = []
        for (x, weight, y, mean, var, gamma, beta), (pad, relu, eps, has_bias, p_local, p_share, up4), gout in zip(mem, ctx.cfg, gouts):
            if gout.dtype != y.dtype:
                gout = to_compute(gout, y.dtype)
            gs.append(gout)
            sums.append(ops.bn_bwd_reduce(y, gout, mean, var, gamma.detach(), beta.detach(), eps, relu))
        packed = torch.cat([t for pair in sums for t in pair])
        dist.all_reduce(packed, group=ctx.sync_group or None)
        SYNC_MESSAGES[1] += 1
        grads, off = [], 0
        for i, ((x, weight, y, mean, var, gamma, beta), (pad, relu, eps, has_bias, p_local, p_share, up4), gout, (dgamma, dbeta)) in \
                enumerate(zip(mem, ctx.cfg, gs, sums)):
            c = weight.shape[0]
            need = ctx.needs_input_grad[2 + 7 * i:2 + 7 * i + 7]
            if ctx.fused_message:
                # p_share holds the GLOBAL count (a device scalar: the count entry of the forward message): the backward-dx kernel
                # divides the all-reduced sums by it itself -- no scaling launches, the message slices are used in place
                dy = ops.bn_bwd_dx_sync(y, gout, mean, var, gamma.detach(), beta.detach(), eps, relu, packed[off:off + c],
                                        packed[off + c:off + 2 * c], p_share.reshape(1), out=y)
                dx, dw = _cba_grads(x, weight, dy, pad, up4, need[0], need[1])
            else:
                sg, sb = packed[off:off + c] * p_share, packed[off + c:off + 2 * c] * p_share
                dx, dw = _bn_bwd_and_grads(x, weight, y, gout, mean, var, gamma.detach(), beta.detach(), eps, relu, sg.contiguous(),
                                           sb.contiguous(), p_local, pad, up4, need[0], need[1])
            off += 2 * c
            dbias = torch.zeros(c, device=x.device, dtype=torch.float32) if has_bias and need[2] else None
            grads += [dx, dw, dbias, dgamma, dbeta, None, None]
        return (None, None, *grads)


def conv_bn_act_group(items: list[dict]) -> list[Tensor]:
    """[conv_bn_act(**item) for item in items] for INDEPENDENT ConvModules (item: x, conv, norm, relu, up).  Under
    SyncBatchNorm training with more than one rank the members share one statistics message per direction
    (_ConvBNActGroupTrain); otherwise they simply run one after the other."""
    norms = [it["norm"] for it in items]
    grouped = (GROUP_SYNC_BN and len(items) > 1 and all(isinstance(nm, nn.SyncBatchNorm) and nm.training for nm in norms)
               and len({id(nm.process_group) for nm in norms}) == 1 and _world(norms[0].process_group) > 1)
    if grouped:
        for it in items:      # members the single-module path would have re-routed (odd shapes) keep that path
            conv, x, up = it["conv"], it["x"], int(it.get("up", 1))
            if up > 1 and not (conv.kernel_size[0] == 3 and conv.padding[0] == 1 and x.shape[1] >= 2 and x.shape[2] >= 2
                               and FUSE_UP4 and conv.weight.shape[0] % 8 == 0 and x.shape[0] * x.shape[1] <= 65535 and up in (2, 4)):
                grouped = False
    if not grouped:
        return [conv_bn_act(it["x"], it["conv"], it["norm"], relu=it.get("relu", True), up=it.get("up", 1)) for it in items]
    metas, flat = [], []
    for it in items:
        conv, norm = it["conv"], it["norm"]
        momentum = 0.1 if norm.momentum is None else norm.momentum
        metas.append((momentum, norm.eps, conv.padding[0], it.get("relu", True), int(it.get("up", 1)) if it.get("up", 1) > 1 else 0))
        flat += [it["x"], conv.weight, conv.bias, norm.weight, norm.bias, norm.running_mean, norm.running_var]
    outs = _ConvBNActGroupTrain.apply(tuple(metas), norms[0].process_group, *flat)
    for norm in norms:
        bump(norm.num_batches_tracked)
    return list(outs)


GROUP_SYNC_BN = True   # A/B switch: False = one statistics exchange per ConvModule (round 2)


FUSE_CONCAT_BWD = True   # A/B switch: False = keep the concat buffer and run the full-resolution data / weight gradients


class _ConcatResizeConvBNTrain(Function):
    """conv3x3(pad 1)(cat([l_0, bilinear(l_1 -> size), ..., bilinear(l_k -> size)])) -> BatchNorm(batch stats) -> ReLU
    (UperNet `fpn_bottleneck` over the upsampled FPN levels, upernet.py:144-152).  Forward: the levels are resized straight
    into the concat buffer and ONE K = 9 * sum(C_j) convolution runs on it; the buffer is a temporary.  Backward, per level:
    the native-resolution level gets its slice of the ordinary data / weight gradient; an upsampled level gets both as GEMMs
    over ITS OWN pixels from the nine gathered maps G_t = resize^T shift_t^T dy (ops.resize_conv3x3_bwd_gather) -- 1/4,
    1/16, 1/64 of the MACs, no bilinear backward, and neither the concat buffer nor its gradient exists in backward."""

    @staticmethod
    def forward(ctx, weight, gamma, beta, running_mean, running_var, momentum, eps, relu, sync_group, *levels):
        cd = levels[0].dtype
        n = weight.shape[0]
        B, H, W, _ = levels[0].shape
        chans = [lv.shape[3] for lv in levels]
        if _tapsum_levels_ok(levels):
            # per level (the convolution is linear over the concat): the native level runs the 3x3 kernel on its own
            # channel slice; every upsampled level contributes its nine tap products, computed at ITS resolution, through
            # one gather-sum pass whose result enters the native level's GEMM as the residual operand
            offs = [sum(chans[:j]) for j in range(len(chans))]
            zs = [ops.conv_gemm(lv, tap_weight(weight, cd, o, o + c)) for lv, o, c in zip(levels[1:], offs[1:], chans[1:])]
            y = ops.conv_gemm(levels[0], slice_weight(weight, cd, 0, chans[0]), R=3, S=3, pad=1,
                              resid=ops.resize_conv3x3_fwd_sum(zs, (H, W)))
            del zs
        else:
            cat = torch.empty((B, H, W, sum(chans)), device=levels[0].device, dtype=cd)
            off = 0
            for lv, c in zip(levels, chans):
                ops.bilinear(lv, (H, W), out=cat[..., off:off + c])
                off += c
            y = ops.conv_gemm(cat, gemm_weight(weight, cd), R=3, S=3, pad=1)
            del cat
        mean, var, world, p_local, p_share = _bn_train_stats(y, n, running_mean, running_var, momentum, sync_group)
        out = ops.bn_apply(y, mean, var, gamma.detach(), beta.detach(), eps, relu)
        ctx.save_for_backward(weight, y, mean, var, gamma, beta, *levels)
        ctx.cfg = (relu, eps, sync_group, world, p_local, p_share, chans)
        return out

    @staticmethod
    def backward(ctx, gout):
        weight, y, mean, var, gamma, beta, *levels = ctx.saved_tensors
        relu, eps, sync_group, world, p_local, p_share, chans = ctx.cfg
        n, ctot = weight.shape[0], weight.shape[1]
        cd = y.dtype
        dy, dgamma, dbeta = _bn_train_backward(y, gout, mean, var, gamma.detach(), beta.detach(), eps, relu, sync_group,
                                               world, p_local, p_share)
        H, W = dy.shape[1], dy.shape[2]
        wd = dgrad_weight(weight, cd)                              # [sum(C_j), 9 N]: the rows of a level are contiguous
        want_dw = ctx.needs_input_grad[0]
        dw = torch.empty((n, 9, ctot), device=y.device, dtype=torch.float32) if want_dw else None
        dls, off = [], 0
        for j, (lv, c) in enumerate(zip(levels, chans)):
            need_dx = ctx.needs_input_grad[9 + j]
            if (lv.shape[1], lv.shape[2]) == (H, W):
                dls.append(ops.conv_gemm(dy, wd[off:off + c], R=3, S=3, pad=1) if need_dx else None)
                if want_dw:
                    dw[:, :, off:off + c] = ops.conv_wgrad(lv, dy, R=3, S=3, pad=1).view(n, 9, c)
            else:
                dx, dwl = ops.resize_conv3x3_bwd(lv, dy, wd[off:off + c] if need_dx else None, want_dw=want_dw)
                dls.append(dx)
                if want_dw:
                    dw[:, :, off:off + c] = dwl.view(n, 9, c)
            off += c
        if want_dw:
            dw = dw.view(n, 3, 3, ctot).permute(0, 3, 1, 2)
        return (dw, dgamma, dbeta, None, None, None, None, None, None, *dls)


def _tapsum_levels_ok(levels) -> bool:
    """levels[0] native, 1..3 further levels each upsampled: by an integer factor of 2 / 4 / 8 (one pass for all of them) or by
    any other ratio up to 10 (one plain gather pass each) -- ops.resize_conv3x3_fwd_sum."""
    B, H, W, _ = levels[0].shape
    return (FUSE_TAPSUM and 2 <= len(levels) <= 4
            and all(ops.resize_conv3x3_fwd_ok((lv.shape[1], lv.shape[2]), (H, W), B)
                    or ops.resize_conv3x3_any_ok((lv.shape[1], lv.shape[2]), (H, W), B) for lv in levels[1:]))


def concat_resize_conv_bn_act(levels: list[Tensor], conv: nn.Conv2d, norm: nn.Module, *, relu: bool = True) -> Tensor:
    """ConvModule(3x3, pad 1, no bias) over cat([levels[0]] + [bilinear(l -> levels[0]'s size) for l in levels[1:]]) on NHWC
    maps.  Training: see _ConcatResizeConvBNTrain; eval (and resize factors above 8): concat_upsample + conv_bn_act."""
    size = (levels[0].shape[1], levels[0].shape[2])
    def factor_ok(lv):
        fy, fx = size[0] / lv.shape[1], size[1] / lv.shape[2]
        return (lv.shape[1], lv.shape[2]) == size or (1.0 < fy <= ops.MAX_ANY_RESIZE and 1.0 < fx <= ops.MAX_ANY_RESIZE)
    ok = (FUSE_CONCAT_BWD and norm.training and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.bias is None
          and conv.weight.shape[0] % 8 == 0 and all(lv.is_contiguous() and factor_ok(lv) for lv in levels)
          and levels[0].shape[0] * max(lv.shape[1] for lv in levels) <= 65535)
    if (not norm.training and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.bias is None
            and all(lv.is_contiguous() for lv in levels) and _tapsum_levels_ok(levels)
            and not (torch.is_grad_enabled() and (conv.weight.requires_grad or any(lv.requires_grad for lv in levels)))):
        # eval, per level: folded BatchNorm scale in every weight slice, epilogue relu(acc + shift + gather-sum)
        cd = levels[0].dtype
        chans = [lv.shape[3] for lv in levels]

        def fold():
            scale, shift = ops.bn_fold(norm.weight.detach(), norm.bias.detach(), norm.running_mean, norm.running_var, norm.eps)
            n, c = conv.weight.shape[0], conv.weight.shape[1]
            w = conv_weight_matrix(conv.weight).float().view(n, 9, c) * scale[:, None, None]
            cast = (lambda t: t) if cd == torch.float32 else (lambda t: ops.cast(t, cd))
            w0 = cast(w[:, :, :chans[0]].reshape(n, -1).contiguous())
            taps, off = [], chans[0]
            for cj in chans[1:]:
                taps.append(cast(w[:, :, off:off + cj].permute(1, 0, 2).reshape(9 * n, cj).contiguous()))
                off += cj
            return w0, taps, shift
        w0, taps, shift = cached((conv.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var),
                                 f"catfold:{cd}:{chans}", fold)
        zs = [ops.conv_gemm(lv, wt) for lv, wt in zip(levels[1:], taps)]
        return ops.conv_gemm(levels[0], w0, R=3, S=3, pad=1, bias=shift, resid=ops.resize_conv3x3_fwd_sum(zs, size),
                             act=ACT_RESID_RELU if relu else ACT_NONE)
    if not ok:
        if FUSE_CONCAT_BWD and norm.training:
            warn_unfused("3x3 ConvModule over a concat of resized levels", f"levels {[tuple(lv.shape) for lv in levels]}: needs "
                         "contiguous levels, resize factors <= 10, N % 8 == 0 and batch * rows <= 65535")
        return conv_bn_act(concat_upsample(levels, size), conv, norm, relu=relu)
    sync_group = norm.process_group if isinstance(norm, nn.SyncBatchNorm) else False
    momentum = 0.1 if norm.momentum is None else norm.momentum
    out = _ConcatResizeConvBNTrain.apply(conv.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var, momentum,
                                         norm.eps, relu, sync_group, *levels)
    bump(norm.num_batches_tracked)
    return out


FUSE_PYRAMID = True   # A/B switch: False = upsample every level into a concat buffer and run one wide 1x1 convolution


def _bn_train_stats(y, n, running_mean, running_var, momentum, sync_group):
    """Batch statistics of a conv output (+ cross-rank merge and the running-stat update): (mean, var, world, p_local, p_share)."""
    world = _world(sync_group) if sync_group is not False else 1
    p_local, p_share = y.numel() // n, None
    if world > 1:
        mean, var = ops.bn_stats(y)
        mean, var, total = sync_batch_stats(mean, var, sync_group or None, count=p_local)
        if running_mean is not None:
            update_running_stats(running_mean, running_var, mean, var, momentum, total)
        p_share = p_local / total
    else:
        mean, var = ops.bn_stats(y, running_mean, running_var, momentum)
        if running_mean is not None:
            mark_updated(running_mean)
            mark_updated(running_var)
    return mean, var, world, p_local, p_share


def _bn_train_backward(y, gout, mean, var, g, b, eps, relu, sync_group, world, p_local, p_share):
    """(dy written over y, dgamma, dbeta) of BatchNorm(batch stats) (+ ReLU)."""
    if gout.dtype != y.dtype:
        gout = to_compute(gout, y.dtype)
    dgamma, dbeta = ops.bn_bwd_reduce(y, gout, mean, var, g, b, eps, relu)
    sg, sb = dgamma, dbeta
    if world > 1:
        sg, sb = sync_sum_pair(dgamma, dbeta, sync_group or None)
        sg, sb = sg * p_share, sb * p_share
    dy = ops.bn_bwd_dx(y, gout, mean, var, g, b, eps, relu, sg, sb, p_local, out=y)
    return dy, dgamma, dbeta


class _PyramidFuseBNTrain(Function):
    """conv1x1(cat([bilinear(l_0 -> size), ..., bilinear(l_{k-1} -> size), l_k])) -> BatchNorm(batch stats) -> ReLU
    (segformer_mlp.py:97-125 `linear_fuse`), evaluated PER LEVEL: a 1x1 convolution acts per pixel and a bilinear resize
    per channel, so they commute -- W_j is applied to level j at the level's own resolution (1/4, 1/16, 1/64 of the
    pixels), the small results are upsampled and summed in one pass (ops.bilinear_sum) and enter the last level's GEMM as
    its residual operand.  Neither the upsampled levels nor the concat buffer exist; forward, data-gradient and
    weight-gradient GEMMs shrink from K = sum(E_j) at full resolution to one K = E_k GEMM at full resolution plus small
    ones.  Backward: dz_j = bilinear^T(dy) per level, then the level's own dgrad / wgrad."""

    @staticmethod
    def forward(ctx, weight, gamma, beta, running_mean, running_var, momentum, eps, relu, sync_group, *levels):
        cd = levels[0].dtype
        n = weight.shape[0]
        wq = gemm_weight(weight, cd)
        size = (levels[-1].shape[1], levels[-1].shape[2])
        offs, off = [], 0
        for lv in levels:
            offs.append(off)
            off += lv.shape[3]
        zs = [ops.conv_gemm(lv, wq[:, o:o + lv.shape[3]]) for lv, o in zip(levels[:-1], offs[:-1])]
        y = ops.conv_gemm(levels[-1], wq[:, offs[-1]:], resid=ops.bilinear_sum(zs, size))
        mean, var, world, p_local, p_share = _bn_train_stats(y, n, running_mean, running_var, momentum, sync_group)
        out = ops.bn_apply(y, mean, var, gamma.detach(), beta.detach(), eps, relu)
        ctx.save_for_backward(weight, y, mean, var, gamma, beta, *levels)
        ctx.cfg = (relu, eps, sync_group, world, p_local, p_share, offs)
        return out

    @staticmethod
    def backward(ctx, gout):
        weight, y, mean, var, gamma, beta, *levels = ctx.saved_tensors
        relu, eps, sync_group, world, p_local, p_share, offs = ctx.cfg
        n, ctot = weight.shape[0], weight.shape[1]
        cd = y.dtype
        dy, dgamma, dbeta = _bn_train_backward(y, gout, mean, var, gamma.detach(), beta.detach(), eps, relu, sync_group,
                                               world, p_local, p_share)
        dzs = [ops.bilinear_bwd(dy, (lv.shape[1], lv.shape[2])) for lv in levels[:-1]] + [dy]
        wd = dgrad_weight(weight, cd)                       # [sum(E_j), N]: rows of level j are contiguous
        dls = []
        for j, (lv, dz, o) in enumerate(zip(levels, dzs, offs)):
            dls.append(ops.conv_gemm(dz, wd[o:o + lv.shape[3]]) if ctx.needs_input_grad[9 + j] else None)
        dw = None
        if ctx.needs_input_grad[0]:
            dw = torch.empty((n, ctot), device=y.device, dtype=torch.float32)
            for lv, dz, o in zip(levels, dzs, offs):
                ops.conv_wgrad(lv, dz, R=1, S=1, dw=dw[:, o:o + lv.shape[3]])
            dw = dw.view(n, ctot, 1, 1)
        return (dw, dgamma, dbeta, None, None, None, None, None, None, *dls)


def pyramid_fuse_bn_act(levels: list[Tensor], conv: nn.Conv2d, norm: nn.Module, *, relu: bool = True) -> Tensor:
    """ConvModule(1x1, no bias) over cat([bilinear(l -> size of levels[-1]) for l in levels[:-1]] + [levels[-1]]) on NHWC
    maps, without the concat (see _PyramidFuseBNTrain).  Eval mode: the folded BN scale goes into the weights, so that every
    level's partial result is already scaled and the last GEMM's epilogue is relu(acc + shift + residual)."""
    ok = (FUSE_PYRAMID and conv.kernel_size == (1, 1) and conv.bias is None and 2 <= len(levels) <= 4
          and all(lv.is_contiguous() for lv in levels) and levels[-1].shape[0] * levels[-1].shape[1] <= 65535)
    size = (levels[-1].shape[1], levels[-1].shape[2])
    if not ok:
        if FUSE_PYRAMID:
            warn_unfused("1x1 ConvModule over a concat of resized levels", f"levels {[tuple(lv.shape) for lv in levels]}: needs 2..4 "
                         "contiguous levels, no bias and batch * rows <= 65535")
        return conv_bn_act(concat_upsample(levels, size), conv, norm, relu=relu)
    if norm.training:
        sync_group = norm.process_group if isinstance(norm, nn.SyncBatchNorm) else False
        momentum = 0.1 if norm.momentum is None else norm.momentum
        out = _PyramidFuseBNTrain.apply(conv.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var,
                                        momentum, norm.eps, relu, sync_group, *levels)
        bump(norm.num_batches_tracked)
        return out
    if torch.is_grad_enabled() and (conv.weight.requires_grad or any(lv.requires_grad for lv in levels)):
        msg = ("gdlhip: autograd through eval-mode BatchNorm is not implemented; call under "
               "torch.no_grad() for inference or model.train() for training")
        raise NotImplementedError(msg)
    cd = levels[0].dtype

    def fold():
        scale, shift = ops.bn_fold(norm.weight.detach(), norm.bias.detach(), norm.running_mean, norm.running_var, norm.eps)
        w = conv_weight_matrix(conv.weight).float() * scale[:, None]
        return (w.contiguous() if cd == torch.float32 else ops.cast(w.contiguous(), cd)), shift
    wq, shift = cached((conv.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var), f"pyrfold:{cd}", fold)
    off, zs = 0, []
    for lv in levels[:-1]:
        zs.append(ops.conv_gemm(lv, wq[:, off:off + lv.shape[3]]))
        off += lv.shape[3]
    return ops.conv_gemm(levels[-1], wq[:, off:], bias=shift, resid=ops.bilinear_sum(zs, size),
                         act=ACT_RESID_RELU if relu else ACT_NONE)


def subpix4_weight(weight: Tensor, cd: torch.dtype) -> dict:
    """Phase weight sets of conv3x3(bilinear_x4(.)) for a [N,C,3,3] conv parameter (cached per parameter version)."""
    def build():
        w = conv_weight_matrix(weight)
        return ops.subpix4_weights(w.float().contiguous(), weight.shape[1], cd)
    return cached((weight,), f"subpix4:{cd}", build)


def conv_bn_act(x: Tensor, conv: nn.Conv2d, norm: nn.Module, *, relu: bool = True, up4: bool = False, up: int = 1) -> Tensor:
    """ConvModule forward on an NHWC tensor (in the compute dtype); returns NHWC.  ``up`` (2 or 4; ``up4`` = 4): the input is
    bilinearly upsampled by that factor first (MultiLevelNeck's fine levels).  Factor 4 runs as sub-pixel phase convolutions
    on the low-resolution map (ops.up4_conv3x3); for both factors the BACKWARD runs at low resolution
    (ops.resize_conv3x3_bwd), the upsampled map is never kept."""
    r = conv.kernel_size[0]
    pad = conv.padding[0]
    up4 = 4 if up4 else (int(up) if up in (2, 4) else 0)      # from here on: the resize factor fused into the node (0 = none)
    if up > 1 and not up4:
        x = bilinear(x, (int(up) * x.shape[1], int(up) * x.shape[2]))
    if up4 and not (r == 3 and pad == 1 and x.shape[1] >= 2 and x.shape[2] >= 2 and FUSE_UP4 and conv.weight.shape[0] % 8 == 0
                    and x.shape[0] * x.shape[1] <= 65535):      # (the gather kernels put batch x rows in one grid dimension)
        if FUSE_UP4:
            warn_unfused(f"ConvModule on a x{up4} resized input", f"input {tuple(x.shape)}, {conv.weight.shape[0]} output channels: "
                         "needs a 3x3 / pad 1 filter, N % 8 == 0 and batch * rows <= 65535")
        x, up4 = bilinear(x, (up4 * x.shape[1], up4 * x.shape[2])), 0
    training = norm.training
    if training:
        if isinstance(norm, nn.SyncBatchNorm):
            sync_group = norm.process_group  # None -> default group
        else:
            sync_group = False
        momentum = 0.1 if norm.momentum is None else norm.momentum
        out = _ConvBNActTrain.apply(x, conv.weight, conv.bias, norm.weight, norm.bias,
                                    norm.running_mean, norm.running_var, momentum, norm.eps, pad,
                                    relu, sync_group, up4)
        bump(norm.num_batches_tracked)
        return out
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        msg = ("gdlhip: autograd through eval-mode BatchNorm is not implemented; call under "
               "torch.no_grad() for inference or model.train() for training")
        raise NotImplementedError(msg)
    cd = x.dtype
    scale, shift = cached((norm.weight, norm.bias, norm.running_mean, norm.running_var), "bnfold",
                          lambda: ops.bn_fold(norm.weight.detach(), norm.bias.detach(),
                                              norm.running_mean, norm.running_var, norm.eps))
    if up4 and FUSE_TAPSUM and ops.resize_conv3x3_fwd_ok((x.shape[1], x.shape[2]), (up4 * x.shape[1], up4 * x.shape[2]),
                                                         x.shape[0]):
        # eval: the BatchNorm scale is folded into the tap weights, shift (+ scale * bias) and the ReLU into the gather-sum
        def fold():
            n, c = conv.weight.shape[0], conv.weight.shape[1]
            w = conv_weight_matrix(conv.weight).float().view(n, 9, c) * scale[:, None, None]
            w = w.permute(1, 0, 2).reshape(9 * n, c).contiguous()
            add = shift if conv.bias is None else (shift + scale * conv.bias.detach().float()).contiguous()
            return (w if cd == torch.float32 else ops.cast(w, cd)), add
        keyp = (conv.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var) + (() if conv.bias is None else (conv.bias,))
        wq, add = cached(keyp, f"tapfold:{cd}", fold)
        return ops.resize_conv3x3_fwd_sum([ops.conv_gemm(x, wq)], (up4 * x.shape[1], up4 * x.shape[2]), addvec=add, relu=relu)
    if up4 == 4:
        return ops.up4_conv3x3(x, subpix4_weight(conv.weight, cd), bias=None if conv.bias is None else conv.bias.detach(),
                               scale=scale, shift=shift, act=ACT_RELU if relu else ACT_NONE)
    if up4:
        x = ops.bilinear(x, (up4 * x.shape[1], up4 * x.shape[2]))
    return ops.conv_gemm(x, gemm_weight(conv.weight, cd), R=r, S=r, pad=pad,
                         bias=None if conv.bias is None else conv.bias.detach(), scale=scale,
                         shift=shift, act=ACT_RELU if relu else ACT_NONE)


# ------------------------------------------------------------------ resampling
class _Bilinear(Function):
    @staticmethod
    def forward(ctx, x, size):
        ctx.in_size = (x.shape[1], x.shape[2])
        return ops.bilinear(x, size)

    @staticmethod
    def backward(ctx, g):
        return ops.bilinear_bwd(g, ctx.in_size), None


def bilinear(x: Tensor, size) -> Tensor:
    """F.interpolate(mode='bilinear', align_corners=False) on NHWC."""
    size = (int(size[0]), int(size[1]))
    if size == (x.shape[1], x.shape[2]):
        return x
    return _Bilinear.apply(x, size)


class _UpsampleAdd(Function):
    """a + bilinear(b -> a's size)  (UperNet top-down path, upernet.py:127-135)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.b_size = (b.shape[1], b.shape[2])
        if a.dim() == 4 and b.dim() == 4 and a.dtype == b.dtype:
            return ops.bilinear_add(a, b)          # one pass (bf16, factor 2 / 4), else copy + accumulate inside the library
        out = ops.copy_cast(a, out=torch.empty(a.shape, device=a.device, dtype=a.dtype))
        return ops.bilinear(b, (a.shape[1], a.shape[2]), out=out, accumulate=True)

    @staticmethod
    def backward(ctx, g):
        return g, ops.bilinear_bwd(g, ctx.b_size)


def upsample_add(a: Tensor, b: Tensor) -> Tensor:
    return _UpsampleAdd.apply(a, b)


class _ConcatUpsample(Function):
    """cat([bilinear(x_i -> size) for x_i], channel dim) written straight into one NHWC buffer
    (upernet.py:103-109,144-152: no separate concat copy)."""

    @staticmethod
    def forward(ctx, size, *xs):
        b = xs[0].shape[0]
        chans = [x.shape[3] for x in xs]
        out = torch.empty((b, size[0], size[1], sum(chans)), device=xs[0].device, dtype=xs[0].dtype)
        off = 0
        for x, c in zip(xs, chans):
            ops.bilinear(x, size, out=out[..., off:off + c])
            off += c
        ctx.meta = [(x.shape[1], x.shape[2], x.shape[3]) for x in xs]
        return out

    @staticmethod
    def backward(ctx, g):
        grads, off = [], 0
        for i, (h, w, c) in enumerate(ctx.meta):
            if ctx.needs_input_grad[i + 1]:
                grads.append(ops.bilinear_bwd(g[..., off:off + c], (h, w)))
            else:
                grads.append(None)
            off += c
        return (None, *grads)


def concat_upsample(xs, size) -> Tensor:
    return _ConcatUpsample.apply((int(size[0]), int(size[1])), *xs)


class _AdaptivePool(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.in_size = (x.shape[1], x.shape[2])
        return ops.adaptive_avgpool(x, s)

    @staticmethod
    def backward(ctx, g):
        return ops.adaptive_avgpool_bwd(g.contiguous(), ctx.in_size), None


def adaptive_avgpool(x: Tensor, s: int) -> Tensor:
    return _AdaptivePool.apply(x, s)


# ------------------------------------------------------------------ classifier tail
class _HeadLogits(Function):
    """1x1 conv to num_classes + bilinear to the input size -> NCHW f32 logits
    (segmentation_head.py:22-26 + dofa.py:89-96; fcn_head.py:69-84 with Dropout2d scale)."""

    @staticmethod
    def forward(ctx, feat, weight, bias, chan_scale, size):
        low = ops.head_1x1(feat, weight.detach(), None if bias is None else bias.detach(), chan_scale)
        ctx.save_for_backward(feat, weight, chan_scale)
        ctx.low_size = (feat.shape[1], feat.shape[2])
        ctx.has_bias = bias is not None
        return ops.upsample_logits(low, size)

    @staticmethod
    def backward(ctx, g):
        feat, weight, chan_scale = ctx.saved_tensors
        dlow = ops.upsample_logits_bwd(g.contiguous(), ctx.low_size)
        dfeat, dw, db = ops.head_1x1_bwd(feat, dlow, weight.detach(), chan_scale,
                                         need_dfeat=ctx.needs_input_grad[0])
        return dfeat, dw.view(weight.shape), (db if ctx.has_bias else None), None, None


class _Head1x1(Function):
    """The classifier alone: NHWC features -> [B, h, w, K] f32 logits at the feature resolution (the first half of _HeadLogits)."""

    @staticmethod
    def forward(ctx, feat, weight, bias, chan_scale):
        low = ops.head_1x1(feat, weight.detach(), None if bias is None else bias.detach(), chan_scale)
        ctx.save_for_backward(feat, weight, chan_scale)
        ctx.has_bias = bias is not None
        return low

    @staticmethod
    def backward(ctx, dlow):
        feat, weight, chan_scale = ctx.saved_tensors
        dfeat, dw, db = ops.head_1x1_bwd(feat, dlow.contiguous(), weight.detach(), chan_scale, need_dfeat=ctx.needs_input_grad[0])
        return dfeat, dw.view(weight.shape), (db if ctx.has_bias else None), None


class LowresLogits(NamedTuple):
    """Logits that have NOT been resized to the image yet: ``low`` [B, h, w, K] f32 (NHWC, as the 1x1 head writes them) and the
    ``size`` the reference's final ``F.interpolate(..., mode="bilinear")`` (dofa.py:89-105) would give them.  What a training
    step hands to ``gdlhip.nn.DiceLoss`` instead of the [B, K, H, W] tensor (round 5: the loss and its gradient are evaluated
    from ``low`` directly; the 168 MB of full-resolution logits per head are never written or read)."""

    low: Tensor
    size: tuple

    def materialise(self) -> Tensor:
        """The [B, K, H, W] f32 logits the reference's forward returns (differentiable)."""
        return _UpsampleLogits.apply(self.low, (int(self.size[0]), int(self.size[1])))


class _UpsampleLogits(Function):
    @staticmethod
    def forward(ctx, low, size):
        ctx.low_size = (low.shape[1], low.shape[2])
        return ops.upsample_logits(low, size)

    @staticmethod
    def backward(ctx, g):
        return ops.upsample_logits_bwd(g.contiguous(), ctx.low_size), None


FUSE_LOWRES_DICE = os.environ.get("GDL_LOWRES_DICE", "1") != "0"   # A/B switch: 0 = training steps materialise the full-resolution logits


def head_logits(feat: Tensor, conv: nn.Conv2d, size, chan_scale: Tensor | None = None, lowres: bool = False):
    """1x1 classifier + bilinear resize to ``size``: NCHW f32 logits; ``lowres``: the classifier's own map and the target size
    (LowresLogits) for a loss that does not need the resized tensor."""
    if lowres:
        return LowresLogits(_Head1x1.apply(feat, conv.weight, conv.bias, chan_scale), (int(size[0]), int(size[1])))
    return _HeadLogits.apply(feat, conv.weight, conv.bias, chan_scale, (int(size[0]), int(size[1])))


class _DiceLoss(Function):
    """smp DiceLoss(mode='multiclass') (configs/dofa_config_RGB.yaml:58-61)."""

    @staticmethod
    def forward(ctx, logits, target, eps):
        loss, sums = ops.dice_loss_fwd(logits, target, eps)
        ctx.save_for_backward(logits, target, sums)
        ctx.eps = eps
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, sums = ctx.saved_tensors
        return ops.dice_loss_bwd(logits, target, sums, g.contiguous().float(), 1.0, ctx.eps), None, None


class _DiceLowres(Function):
    """smp DiceLoss(mode='multiclass') of bilinear(low -> size), forward and backward from the low-resolution map."""

    @staticmethod
    def forward(ctx, low, target, size, eps):
        loss, sums = ops.dice_loss_lowres_fwd(low, target, size, eps)
        ctx.save_for_backward(low, target, sums)
        ctx.size, ctx.eps = size, eps
        return loss

    @staticmethod
    def backward(ctx, g):
        low, target, sums = ctx.saved_tensors
        return ops.dice_loss_lowres_bwd(low, target, ctx.size, sums, g.contiguous().float(), 1.0, ctx.eps), None, None, None


class _DiceBinaryLoss(Function):
    """smp DiceLoss(mode='binary') (configs/unetplus_config_RGB.yaml:40-47, num_classes 1)."""

    @staticmethod
    def forward(ctx, logits, target, eps):
        loss, sums = ops.dice_binary_loss_fwd(logits, target, eps)
        ctx.save_for_backward(logits, target, sums)
        ctx.eps = eps
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, sums = ctx.saved_tensors
        return ops.dice_binary_loss_bwd(logits, target, sums, g.contiguous().float(), 1.0, ctx.eps), None, None


class DiceLoss(nn.Module):
    """Drop-in for ``segmentation_models_pytorch.losses.DiceLoss`` in the two modes the reference's configs use:
    ``mode="multiclass"`` (configs/dofa_config_RGB.yaml:58-61, segformer) and ``mode="binary"``
    (configs/unetplus_config_RGB.yaml, ``num_classes: 1``); ``smooth=0``, ``ignore_index=None``, ``from_logits=True``,
    ``log_loss=False``, ``classes=None`` (smp's defaults, which are also what the configs pass)."""

    def __init__(self, mode: str = "multiclass", classes=None, log_loss: bool = False, from_logits: bool = True,
                 smooth: float = 0.0, ignore_index=None, eps: float = 1e-7) -> None:
        super().__init__()
        if mode not in ("multiclass", "binary"):
            msg = f"gdlhip DiceLoss implements mode='multiclass' and mode='binary' (got {mode!r})"
            raise NotImplementedError(msg)
        if classes is not None or log_loss or not from_logits or smooth != 0.0 or ignore_index is not None:
            msg = ("gdlhip DiceLoss implements smp's defaults (classes=None, log_loss=False, from_logits=True, "
                   "smooth=0, ignore_index=None), which are what the reference's configs use")
            raise NotImplementedError(msg)
        self.mode, self.eps = mode, eps

    def forward(self, y_pred, y_true: Tensor) -> Tensor:
        if isinstance(y_pred, LowresLogits):
            # a training step's not-yet-resized logits: the loss (and its gradient) straight from the low-resolution map
            size = (int(y_pred.size[0]), int(y_pred.size[1]))
            yt = y_true[:, 0] if y_true.dim() == 4 and y_true.shape[1] == 1 else y_true
            if (self.mode == "multiclass" and FUSE_LOWRES_DICE and ops.dice_lowres_ok(y_pred.low, size)
                    and y_pred.low.dtype == torch.float32 and tuple(yt.shape[1:]) == size):
                return _DiceLowres.apply(y_pred.low.contiguous(), yt.long().contiguous(), size, self.eps)
            y_pred = y_pred.materialise()
        if y_pred.dtype != torch.float32 or not y_pred.is_contiguous():
            y_pred = y_pred.float().contiguous()
        if self.mode == "binary":
            if y_pred.shape[0] != y_true.shape[0] or y_pred.numel() != y_true.numel():
                msg = f"DiceLoss(binary): y_pred {tuple(y_pred.shape)} and y_true {tuple(y_true.shape)} do not match"
                raise ValueError(msg)
            return _DiceBinaryLoss.apply(y_pred, y_true.long().contiguous(), self.eps)
        if y_true.dim() == y_pred.dim() and y_true.shape[1] == 1:
            y_true = y_true[:, 0]          # smp views the target as [B, -1]: an un-squeezed [B,1,H,W] mask is the same
        return _DiceLoss.apply(y_pred, y_true.long().contiguous(), self.eps)


def predict_mask(logits) -> Tensor:
    """``softmax(dim=1).argmax(dim=1)`` (segmentation_dofa.py:281).  For not-yet-resized logits (LowresLogits) the resize is
    evaluated per pixel inside the kernel: the same mask without the [B, K, H, W] tensor."""
    if isinstance(logits, LowresLogits):
        low = logits.low
        if low.dtype == torch.float32 and low.dim() == 4 and 2 <= low.shape[3] <= 16:
            return ops.upsample_argmax(low.contiguous(), logits.size)
        logits = logits.materialise()
    return ops.softmax_argmax(logits.float().contiguous())


# ------------------------------------------------------------------ optimizer
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics with the update (and optional global-norm clipping, Lightning's
    ``gradient_clip_val``) done by HIP kernels.  Reference: configs/dofa_config_RGB.yaml:11,62-65."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, max_grad_norm: float | None = None, capturable: bool = False) -> None:
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        # capturable (torch.optim.Adam(capturable=True)): step count, bias corrections and hyper-parameters live in device
        # memory, so that a hipGraph-captured step (gdlhip.graphs) replays correctly; chunk tables come from pinned memory
        self.capturable = capturable
        self._dev_state: dict = {}    # param group index -> f32[8] {step, lr, b1, b2, eps, wd, bc1, bc2}
        self._dev_lr: dict = {}       # param group index -> the learning rate last written to the device state
        self._table_bufs: dict = {}   # param group index -> (pinned, device) chunk-table buffers, allocated once
        self._acc = None
        self._tables: dict = {}      # param group index -> (address signature, device chunk table)
        self.table_builds = 0        # how often a chunk table was (re)built: 1 per group in steady state
        # bf16 GEMM operands of the parameters (gemm_weight cache) are rewritten by the update kernel itself
        self.shadows = LAUNCH_FUSION
        self._shadowed: list = []
        self._repack = None          # (signature, device table, tiles) of the derived-operand rebuild (refresh_derived)
        self._derived_last: list = []   # (cache key, operand, parameter) of the last refresh_derived
        self._repack_scan = None     # ((operand builds so far, registered parameters, updated parameters), entries) of the last scan

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        todo = [(gi, g, p) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.grad is not None]
        if not todo:
            return loss
        dev = todo[0][2].device
        # one chunk table per (param group, step count) -> one launch each.  A table only holds addresses and sizes: it is
        # rebuilt (535 rows for DOFA-base + one H2D copy) only when a parameter, gradient or state buffer moved -- with
        # gradient_as_bucket_view / set_to_none=False and a caching allocator that is the first step only
        buckets: dict = {}
        shadowed: list = []
        for gi, group, p in todo:
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            if p.grad.stride() != p.stride():
                p.grad = _restride(p.grad, p)
            _flat(p), _flat(p.grad)  # layout check (dense storage)
            sig = buckets.setdefault((gi, st["step"]), [])
            sh = gemm_weight_shadow(p) if self.shadows else None
            if sh is not None:
                shadowed.append((sh[0], sh[1], p))     # the tensor is held: the table's address stays valid
            sig.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                        0 if sh is None else sh[1].data_ptr()))
            mark_updated(p)
        tables = {}
        for key, sig in buckets.items():
            sig = tuple(sig)
            hit = self._tables.get(key[0])
            if hit is None or hit[0] != sig:
                rows = [(pp + 4 * off, gp + 4 * off, mp + 4 * off, vp + 4 * off, min(self.CHUNK, n - off), sp + 2 * off if sp else 0)
                        for pp, gp, mp, vp, n, sp in sig for off in range(0, n, self.CHUNK)]
                host = torch.tensor(rows, dtype=torch.int64)
                hit = (sig, self._upload(key[0], host, dev))
                if len(buckets) == len({k[0] for k in buckets}):      # (one step count per group: the usual case -> cacheable)
                    self._tables[key[0]] = hit
                self.table_builds += 1
            tables[key] = hit[1]
        clip = None
        if self.max_grad_norm is not None:
            if self._acc is None:
                self._acc = torch.zeros(2, device=dev, dtype=torch.float32)
            self._acc.zero_()
            for t in tables.values():
                ops.multi_sumsq(t, self._acc[0:1])
            ops.clip_coef(self._acc[0:1], float(self.max_grad_norm), self._acc[1:2])
            clip = self._acc[1:2]
        if self.capturable and not torch.cuda.is_current_stream_capturing():
            self.sync_lr()                # a scheduler's param_groups["lr"] write reaches the device-side hyper-parameters
        for (gi, step), t in tables.items():
            group = self.param_groups[gi]
            if self.capturable:
                ops.adam_tick(self.device_state(gi, dev), *group["betas"])
                ops.multi_adam_dev(t, self._dev_state[gi], clip)
            else:
                ops.multi_adam(t, group["lr"], *group["betas"], group["eps"], group["weight_decay"], step, clip)
        self._shadowed = shadowed
        for key, val, p in shadowed:      # the kernels above rewrote the bf16 GEMM operands too
            refresh_shadow(key, val, p)
        if self.shadows:
            self.refresh_derived([p for _, _, p in todo], dev)
        return loss

    def _upload(self, name, host: Tensor, dev) -> Tensor:
        """A host int64 table on the device.  capturable: under stream capture nothing may be allocated (pinned or device), so
        both buffers exist since the first table of this name and shape; the captured H2D copy re-reads the pinned one at
        every replay."""
        if not self.capturable:
            return host.to(dev, non_blocking=True)
        bufs = self._table_bufs.get(name)
        if bufs is None or bufs["dev"].shape != host.shape:
            pin = lambda: torch.empty(host.shape, dtype=torch.int64).pin_memory()      # noqa: E731
            bufs = {"eager": [pin(), pin()], "events": [None, None], "capture": pin(), "i": 0,
                    "dev": torch.empty(host.shape, dtype=torch.int64, device=dev)}
            self._table_bufs[name] = bufs
        if torch.cuda.is_current_stream_capturing():
            # the captured H2D copy re-reads this pinned buffer at every replay: eager rebuilds never touch it
            src = bufs["capture"]
            src.copy_(host)
            bufs["dev"].copy_(src, non_blocking=True)
        else:
            # eager steps with set_to_none gradients rebuild the table every step; its asynchronous H2D copy reads
            # the pinned buffer later, so the host alternates between two and only waits for the copy issued two
            # rebuilds ago (long done) instead of overwriting a buffer whose copy has not run yet
            i = bufs["i"] = bufs["i"] ^ 1
            if bufs["events"][i] is not None:
                bufs["events"][i].synchronize()
            src = bufs["eager"][i]
            src.copy_(host)
            bufs["dev"].copy_(src, non_blocking=True)
            ev = bufs["events"][i] or torch.cuda.Event()
            ev.record()
            bufs["events"][i] = ev
        return bufs["dev"]

    @torch.no_grad()
    def refresh_derived(self, params=None, dev=None) -> int:
        """Rebuild, in ONE launch (gdl_multi_repack), every bf16 operand that was derived from the given (default: all) conv
        parameters in another element order and sits in the operand cache -- channel slices and tap-major forms of the 3x3
        parameters, data-gradient operands -- and mark the entries current.  Called behind every update; GraphedTrainStep calls
        it after restoring the parameters.  Returns the number of operands rewritten."""
        everything = params is None
        if everything:
            params = [p for g in self.param_groups for p in g["params"]]
        # steady state: nothing was built since the last scan (the entries were rewritten in place and revalidated), the same
        # parameters were updated -> the same entries, no scan
        scan_key = (_CACHE_BUILDS[0], len(_DERIVED), len(params))
        if not everything and self._repack_scan is not None and self._repack_scan[0] == scan_key:
            ents = self._repack_scan[1]
        else:
            ents = [(p, *e) for p in params for e in derived_operands(p)]
            self._repack_scan = None if everything else (scan_key, ents)
        if not ents:
            return 0
        dev = dev or ents[0][0].device
        sig = tuple((p.data_ptr(), val.data_ptr(), mode, c0, c1) for p, _, val, mode, c0, c1 in ents)
        hit = self._repack
        # (inside a stream capture the table is ALWAYS uploaded: the captured copy then restores it from its own pinned buffer
        # at every replay, whatever eager steps in between wrote into the device buffer -- like the update's chunk tables, whose
        # gradient addresses change with every capture)
        if hit is None or hit[0] != sig or (self.capturable and torch.cuda.is_current_stream_capturing()):
            rows, tile0 = [], 0
            for p, _, val, mode, c0, c1 in ents:
                n, c, t = p.shape[0], p.shape[1], p.shape[2] * p.shape[3]
                cs = c1 - c0
                if val.numel() != n * t * cs:
                    msg = f"gdlhip FusedAdam: derived operand {tuple(val.shape)} does not match parameter {tuple(p.shape)}[{c0}:{c1}]"
                    raise ValueError(msg)
                tiles_c = (cs + 31) // 32
                rows.append((p.data_ptr(), val.data_ptr(), n, t, c, c0, cs, mode, tile0, tiles_c))
                tile0 += t * ((n + 31) // 32) * tiles_c
            hit = self._repack = (sig, self._upload("repack", torch.tensor(rows, dtype=torch.int64), dev), tile0)
        ops.multi_repack(hit[1], hit[2])
        for p, key, val, _, _, _ in ents:
            refresh_shadow(key, val, p)
        self._derived_last = [(key, val, p) for p, key, val, _, _, _ in ents]
        return len(ents)

    def device_state(self, gi: int, dev=None) -> Tensor:
        """The device-side {step, lr, b1, b2, eps, wd, bc1, bc2} of param group gi (capturable mode), created from the group's
        current hyper-parameters on first use.  A scheduler's new learning rate reaches a captured step through sync_lr()."""
        st = self._dev_state.get(gi)
        if st is None:
            g = self.param_groups[gi]
            steps = {self.state[p]["step"] - 1 for p in g["params"] if p in self.state and self.state[p]}
            step0 = float(steps.pop()) if len(steps) == 1 else 0.0
            st = torch.tensor([step0, g["lr"], *g["betas"], g["eps"], g["weight_decay"], 0.0, 0.0], dtype=torch.float32).to(dev)
            self._dev_state[gi] = st
            self._dev_lr[gi] = float(g["lr"])
        return st

    def sync_lr(self) -> None:
        """Write the param groups' current learning rates into the device state when a scheduler changed them (one fill_ on the
        stream per changed group, nothing otherwise).  Capturable mode reads lr from the device: called at the start of every
        eager step and by GraphedTrainStep before every replay."""
        for gi, st in self._dev_state.items():
            lr = float(self.param_groups[gi]["lr"])
            if self._dev_lr.get(gi) != lr:
                st[1:2].fill_(lr)
                self._dev_lr[gi] = lr

    def note_replay(self) -> None:
        """A captured step ran: advance the host-side step counts (what state_dict() / checkpoints report) like the device's."""
        for g in self.param_groups:
            for p in g["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] += 1

    CHUNK = 65536


def _flat(t: Tensor) -> Tensor:
    """The dense storage of a (possibly channels_last) tensor, as a flat view."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)
    msg = f"gdlhip FusedAdam: parameter layout {t.stride()} is not dense"
    raise ValueError(msg)


def _restride(g: Tensor, p: Tensor) -> Tensor:
    out = torch.empty_like(p, memory_format=torch.preserve_format)
    out.copy_(g)
    return out
