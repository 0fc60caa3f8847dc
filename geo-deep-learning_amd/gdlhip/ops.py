"""Tensor-level wrappers over the C-ABI (raw ops, no autograd).

Conventions: activations are NHWC torch tensors ``[B, H, W, C]`` (any strides with
``stride(-1) == 1``); 2-D ``[rows, C]`` tensors are treated as ``[1, 1, rows, C]``.
PyTorch only provides device memory and the stream here -- all arithmetic happens in
libgdlhip.so.  Shape / dtype / alignment errors raise ValueError like the reference's own
argument checks (dofa_v2.py:439-441, multilevel_neck.py:141-146).
"""

from __future__ import annotations

import ctypes as C
import os

import torch
from torch import Tensor

from . import _lib
from ._lib import (ACT_GELU, ACT_MUL_GELU_GRAD, ACT_NONE, ACT_RELU, ACT_RESID_RELU, BF16, F32, ConvArgs,  # noqa: F401
                   WgradArgs, check)

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t: Tensor | torch.dtype) -> int:
    d = t if isinstance(t, torch.dtype) else t.dtype
    try:
        return _DT[d]
    except KeyError:
        raise ValueError(f"gdlhip: unsupported dtype {d}") from None


def _p(t: Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("gdlhip ops need CUDA/HIP device tensors (no CPU fallback)")


def _f32vec(t: Tensor | None, n: int, name: str) -> Tensor | None:
    if t is None:
        return None
    if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous f32 vector of {n} elements")
    return t


def image_f32(img: Tensor, who: str) -> Tensor:
    """Model-entry check: tiles must arrive normalised (floating point).  A raw integer tile would silently train on
    0..255 pixels -- the reference never sees one (its workers normalise, wds_dataset.py:230-236); here raw tiles are
    only legal upstream of DeviceInputStage / normalize_raw."""
    if not img.is_floating_point():
        raise TypeError(f"{who}: image dtype {img.dtype} is a raw tile; normalise it first (DeviceInputStage, "
                        f"gdlhip.ops.normalize_raw or the host path of SampleProcessor)")
    return img.float().contiguous()


def as_nhwc(x: Tensor) -> Tensor:
    """Logical NCHW tensor -> NHWC view (copying only if its channels are not contiguous)."""
    v = x.permute(0, 2, 3, 1)
    return v if v.stride(-1) == 1 else v.contiguous()


def as_nchw(x: Tensor) -> Tensor:
    """NHWC tensor -> logical NCHW view (== torch.channels_last memory format)."""
    return x.permute(0, 3, 1, 2)


def _nhwc4(x: Tensor, name: str) -> Tensor:
    if x.dim() == 2:
        x = x.unsqueeze(0).unsqueeze(0)
    if x.dim() != 4 or (x.stride(-1) != 1 and x.shape[-1] != 1):
        raise ValueError(f"{name}: expected NHWC tensor with unit channel stride, got "
                         f"shape {tuple(x.shape)} strides {x.stride()}")
    # strides of size-1 dims are arbitrary in torch: canonicalise them (alignment checks, "dense" tests)
    B, H, W, Cc = x.shape
    sB, sH, sW, _ = x.stride()
    nW = sW if W > 1 else Cc
    nH = sH if H > 1 else W * nW
    nB = sB if B > 1 else H * nH
    if (nB, nH, nW) != (sB, sH, sW) or x.stride(-1) != 1:
        x = x.as_strided((B, H, W, Cc), (nB, nH, nW, 1))
    return x


# ------------------------------------------------------------------ conv / linear (MFMA)
def conv_gemm(x: Tensor, w: Tensor, *, R: int = 1, S: int = 1, stride: int = 1, pad: int = 0,
              bias: Tensor | None = None, scale: Tensor | None = None, shift: Tensor | None = None,
              act: int = ACT_NONE, batch_scale: Tensor | None = None, resid: Tensor | None = None,
              out: Tensor | None = None, out_dtype: torch.dtype | None = None,
              alpha: float = 1.0, aux_out: Tensor | None = None, want_stats: bool = False):
    """out = epilogue(conv(x, w)); x NHWC [B,H,W,C], w [N, R*S*C] (K order r,s,c).
    ``aux_out`` (same shape / dtype / strides as out) receives the pre-activation values.
    ``want_stats``: returns (out, partials, rows) -- when the call can emit them (gdl_conv_gemm_stats_rows: bf16, bias-only
    epilogue, whole tiles) `partials` [rows, 2, N] f32 holds per-channel sums and sums of squares of the bf16 outputs for
    bn_stats_finalize (train-mode BatchNorm statistics without a pass over `out`); otherwise (out, None, 0)."""
    _need_cuda(x, w)
    two_d = x.dim() == 2
    x4 = _nhwc4(x, "conv_gemm input")
    B, H, W, Cc = x4.shape
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != R * S * Cc or w.stride(1) != 1 or w.dtype != x.dtype:
        raise ValueError(f"conv_gemm: weight must be [{N}, {R * S * Cc}] {x.dtype} K-contiguous, "
                         f"got {tuple(w.shape)} {w.dtype}")
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - S) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, N), device=x.device, dtype=out_dtype or x.dtype)
        out4 = out
    else:
        out4 = _nhwc4(out, "conv_gemm out")
        if tuple(out4.shape) != (B, Ho, Wo, N):
            raise ValueError(f"conv_gemm: out shape {tuple(out4.shape)} != {(B, Ho, Wo, N)}")
    a = ConvArgs()
    a.inp, a.dtype = x4.data_ptr(), dt(x)
    a.B, a.H, a.W, a.C = B, H, W, Cc
    a.in_sB, a.in_sH, a.in_sW = x4.stride(0), x4.stride(1), x4.stride(2)
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad = Ho, Wo, R, S, stride, pad
    a.w, a.w_sN, a.N = w.data_ptr(), w.stride(0), N
    a.out, a.out_dtype = out4.data_ptr(), dt(out4)
    a.out_sB, a.out_sH, a.out_sW = out4.stride(0), out4.stride(1), out4.stride(2)
    a.alpha = alpha
    a.bias = None if bias is None else _f32vec(bias, N, "bias").data_ptr()
    a.scale = None if scale is None else _f32vec(scale, N, "scale").data_ptr()
    a.shift = None if shift is None else _f32vec(shift, N, "shift").data_ptr()
    a.act = act
    a.batch_scale = None if batch_scale is None else _f32vec(batch_scale, B, "batch_scale").data_ptr()
    if resid is not None:
        r4 = _nhwc4(resid, "conv_gemm resid")
        if r4.shape[-1] != N or r4.shape[1:3] != out4.shape[1:3]:
            raise ValueError("conv_gemm: resid shape mismatch")
        a.resid, a.resid_dtype = r4.data_ptr(), dt(r4)
        a.res_sB = r4.stride(0) if r4.shape[0] == B else 0
        a.res_sH, a.res_sW = r4.stride(1), r4.stride(2)
    if aux_out is not None:
        x4a = _nhwc4(aux_out, "conv_gemm aux_out")
        if x4a.shape != out4.shape or x4a.stride() != out4.stride() or x4a.dtype != out4.dtype:
            raise ValueError("conv_gemm: aux_out must match out in shape, strides and dtype")
        a.aux_out = x4a.data_ptr()
    a.nz, a.nz_inner = 1, 1
    partials, rows = None, 0
    if want_stats:
        rows = int(_lib.load().gdl_conv_gemm_stats_rows(C.byref(a)))
        if rows:
            partials = torch.empty((rows, 2, N), device=x.device, dtype=torch.float32)
            a.stats_partial = partials.data_ptr()
    _launch_conv_gemm(a, "gdl_conv_gemm")
    if two_d and out4 is out:
        out = out.view(Wo, N)
    return (out, partials, rows) if want_stats else out


class KernelTimer:
    """HIP-event timing of every gdl_conv_gemm launch (bench.py's roofline leg).  Events are
    recorded on the stream the kernels are launched on (torch's current stream)."""

    VARIANT = {0: "conv_gemm_kernel<{dt},2,2,1,1> (64x64 tiles)", 1: "conv_gemm_kernel<{dt},2,2,2,2> (128x128 tiles)",
               2: "conv_gemm_kernel<{dt},2,4,4,2> (256x256 tiles)",
               3: "conv_gemm_kernel<{dt},2,4,4,2,pingpong> (256x256 tiles, alternating loader halves)",
               4: "conv3x3_sf_kernel<{dt}> (256x256 tiles, 3x3 taps share one staged activation tile)",
               5: "conv_gemm_kernel<{dt},4,1,2,2> (256x64 tiles, narrow outputs)",
               6: "conv_gemm_dual_kernel (256x128 tiles, four waves, two workgroups resident per CU)",
               8: "conv_gemm_w4_kernel (256x256 tiles, one wave per SIMD, every load in an MFMA shadow)",
               9: "conv_gemm_persist_kernel (256x256 ping-pong tiles, one persistent workgroup per CU, next tile's first stage under the epilogue)",
               10: "conv_gemm_w4p_kernel (256x256 tiles, one persistent workgroup per CU, one wave per SIMD, finished tile parked in registers and stored from the next tile's MFMA shadows)",
               7: "conv3x3_narrow_kernel (direct 3x3, C <= 32, one staged window per 4 x 64 pixels x 32 output channels)",
               12: "conv_gemm_kernel<{dt},2,2,1,1,stages=4> (64x64 tiles, four LDS stages: few tiles, long K)"}

    def __init__(self) -> None:
        self.records: list = []
        self.variants: list = []      # the planner's tile variant of every launch, in launch order

    def launch(self, a: ConvArgs, what: str) -> None:
        lib = _lib.load()
        fl = C.c_int64()
        variant = lib.gdl_conv_gemm_plan(C.byref(a), C.byref(fl))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        status = lib.gdl_conv_gemm(C.byref(a), _stream())
        e1.record()
        check(status, what)
        key = self.VARIANT[variant].format(dt="bf16" if a.dtype == BF16 else "f32")
        # algorithmic bytes of the launch: every operand read once, the result written once, at their dtypes
        es, oes = (2 if a.dtype == BF16 else 4), (2 if a.out_dtype == BF16 else 4)
        nz = int(a.nz)
        nbytes = nz * (int(a.B) * int(a.H) * int(a.W) * int(a.C) * es + int(a.N) * int(a.R) * int(a.S) * int(a.C) * es
                       + int(a.B) * int(a.Ho) * int(a.Wo) * int(a.N) * oes)
        if a.resid:
            nbytes += nz * int(a.B) * int(a.Ho) * int(a.Wo) * int(a.N) * (2 if a.resid_dtype == BF16 else 4)
        shape = (nz * int(a.B) * int(a.Ho) * int(a.Wo), int(a.N), int(a.R) * int(a.S) * int(a.C), int(a.R), int(a.S), oes, bool(a.resid))
        self.records.append((key, fl.value, e0, e1, int(a.R) * int(a.S) * int(a.C), nbytes, shape))
        self.variants.append(variant)

    @staticmethod
    def _k_bucket(k: int) -> str:
        return "K<=1024" if k <= 1024 else ("1024<K<=4096" if k <= 4096 else "K>4096")

    def summary(self) -> dict:
        """Per kernel class: launches, ms, flops, and the same split by reduction depth K = R*S*C (`by_k`): one class runs
        12-K-step ViT linears and 100-K-step convolutions, whose achievable rates differ by a factor of two."""
        torch.cuda.synchronize()
        out: dict = {}
        for key, flops, e0, e1, kdepth, nbytes, shape in self.records:
            ms = e0.elapsed_time(e1)
            s = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0, "by_k": {}, "by_shape": {}})
            m_, n_, k_, r_, s_, oes, res = shape
            tag = f"M{m_} N{n_} K{k_} {r_}x{s_} -> {'bf16' if oes == 2 else 'f32'}{' +resid' if res else ''}"
            for d in (s, s["by_k"].setdefault(self._k_bucket(kdepth), {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0}),
                      s["by_shape"].setdefault(tag, {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0})):
                d["launches"] += 1
                d["ms"] += ms
                d["flops"] += flops
                d["bytes"] += nbytes
        return out


TIMER: KernelTimer | None = None


def _launch_conv_gemm(a: ConvArgs, what: str) -> None:
    if TIMER is not None:
        TIMER.launch(a, what)
    else:
        check(_lib.load().gdl_conv_gemm(C.byref(a), _stream()), what)


def linear(x: Tensor, w: Tensor, bias: Tensor | None = None, **kw) -> Tensor:
    """y[..., N] = epilogue(x[..., K] @ w[N, K]^T); rows may be strided."""
    K = x.shape[-1]
    lead = x.shape[:-1]
    x2 = x.reshape(-1, K)
    if "resid" in kw and kw["resid"] is not None:
        kw["resid"] = kw["resid"].reshape(-1, w.shape[0])
    for key in ("out", "aux_out"):
        if kw.get(key) is not None:
            kw[key] = kw[key].reshape(-1, w.shape[0])
    y = conv_gemm(x2, w, bias=bias, **kw)
    return y.reshape(*lead, w.shape[0])


def batched_gemm_raw(a: ConvArgs) -> None:
    _launch_conv_gemm(a, "gdl_conv_gemm(batched)")


def conv_gemm_grouped(x: Tensor, w: Tensor, *, R: int, S: int, stride: int = 1, pad: int = 0, out: Tensor | None = None) -> Tensor:
    """Grouped convolution as ONE batched implicit-GEMM launch: x NHWC [B,H,W,Z*c], w [Z, n, R*S*c] (K order r,s,c) -> out
    [B,Ho,Wo,Z*n]; problem z reads channels z*c.. of x with filter w[z] and writes channels z*n.. of out (grid.z = z: the
    operands are channel slices addressed through the batch strides of gdl_conv_args, nothing is copied).  No epilogue terms:
    per-channel vectors are not offset per z by the kernels."""
    _need_cuda(x, w)
    x4 = _nhwc4(x, "conv_gemm_grouped input")
    B, H, W, Ct = x4.shape
    Z, n, kk = w.shape
    c = kk // (R * S)
    if w.dim() != 3 or c * R * S != kk or Z * c != Ct or not w.is_contiguous() or w.dtype != x.dtype:
        raise ValueError(f"conv_gemm_grouped: weight must be contiguous [Z, n, {R * S}*c] {x.dtype} with Z*c = {Ct}, got {tuple(w.shape)} {w.dtype}")
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - S) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, Z * n), device=x.device, dtype=x.dtype)
    o4 = _nhwc4(out, "conv_gemm_grouped out")
    if tuple(o4.shape) != (B, Ho, Wo, Z * n):
        raise ValueError(f"conv_gemm_grouped: out shape {tuple(o4.shape)} != {(B, Ho, Wo, Z * n)}")
    a = ConvArgs()
    a.inp, a.dtype = x4.data_ptr(), dt(x4)
    a.B, a.H, a.W, a.C = B, H, W, c
    a.in_sB, a.in_sH, a.in_sW = x4.stride(0), x4.stride(1), x4.stride(2)
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad = Ho, Wo, R, S, stride, pad
    a.w, a.w_sN, a.N = w.data_ptr(), kk, n
    a.out, a.out_dtype = o4.data_ptr(), dt(o4)
    a.out_sB, a.out_sH, a.out_sW = o4.stride(0), o4.stride(1), o4.stride(2)
    a.alpha, a.act = 1.0, ACT_NONE
    a.nz, a.nz_inner = Z, 1
    a.in_sZ0, a.w_sZ0, a.out_sZ0 = c, n * kk, n
    _launch_conv_gemm(a, "gdl_conv_gemm(grouped)")
    return out


def conv_wgrad_grouped(x: Tensor, dy: Tensor, *, Z: int, R: int, S: int, stride: int = 1, pad: int = 0) -> Tensor:
    """Weight gradient of conv_gemm_grouped: dw [Z, n, R*S*c] f32, problem z = channel slices z*c.. of x and z*n.. of dy."""
    _need_cuda(x, dy)
    x4, dy4 = _nhwc4(x, "wgrad x"), _nhwc4(dy, "wgrad dy")
    if x4.dtype != dy4.dtype:
        raise ValueError("conv_wgrad_grouped: x and dy dtypes differ")
    B, H, W, Ct = x4.shape
    _, Ho, Wo, Nt = dy4.shape
    if Ct % Z or Nt % Z:
        raise ValueError("conv_wgrad_grouped: channel counts must be multiples of Z")
    c, n = Ct // Z, Nt // Z
    dw = torch.empty((Z, n, R * S * c), device=x.device, dtype=torch.float32)
    a = WgradArgs()
    a.inp, a.dy, a.dtype = x4.data_ptr(), dy4.data_ptr(), dt(x4)
    a.B, a.H, a.W, a.C = B, H, W, c
    a.in_sB, a.in_sH, a.in_sW = x4.stride(0), x4.stride(1), x4.stride(2)
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad, a.N = Ho, Wo, R, S, stride, pad, n
    a.dy_sB, a.dy_sH, a.dy_sW = dy4.stride(0), dy4.stride(1), dy4.stride(2)
    a.dw, a.dw_sN, a.accumulate = dw.data_ptr(), R * S * c, 0
    a.nz, a.nz_inner = Z, 1
    a.in_sZ0, a.dy_sZ0, a.dw_sZ0 = c, n, n * R * S * c
    lib = _lib.load()
    nbytes = lib.gdl_conv_wgrad_workspace(C.byref(a))
    ws = torch.empty(max(nbytes, 4) // 4, device=x.device, dtype=torch.float32)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.gdl_conv_wgrad(C.byref(a), _stream()), "gdl_conv_wgrad(grouped)")
    return dw


# (Round 6, measured and removed: weight gradients launched on a SIDE stream -- leaves of the backward pass, MFMA-bound, meant to
# share the chip with the HBM-bound BatchNorm-backward / gather kernels that follow on the main stream; joined by an autograd engine
# callback at the end of backward.  Three same-box A/B repetitions at batch 64: 936.5-944.1 without, 934.7-937.7 with
# (profiles/r06o_*): a GEMM-class workgroup takes a CU's whole register file and LDS, so nothing co-resides with it, and the GEMM
# phases already sit at the power cap.)
def conv_wgrad(x: Tensor, dy: Tensor, *, R: int, S: int, stride: int = 1, pad: int = 0,
               dw: Tensor | None = None, accumulate: bool = False) -> Tensor:
    """dw[N, R*S*C] (f32) = sum_pixels dy[.., n] * x[.. + tap, c]."""
    _need_cuda(x, dy)
    x4, dy4 = _nhwc4(x, "wgrad x"), _nhwc4(dy, "wgrad dy")
    if x4.dtype != dy4.dtype:
        raise ValueError("conv_wgrad: x and dy dtypes differ")
    B, H, W, Cc = x4.shape
    _, Ho, Wo, N = dy4.shape
    if dw is None:
        dw = torch.empty((N, R * S * Cc), device=x.device, dtype=torch.float32)
    a = WgradArgs()
    a.inp, a.dy, a.dtype = x4.data_ptr(), dy4.data_ptr(), dt(x4)
    a.B, a.H, a.W, a.C = B, H, W, Cc
    a.in_sB, a.in_sH, a.in_sW = x4.stride(0), x4.stride(1), x4.stride(2)
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad, a.N = Ho, Wo, R, S, stride, pad, N
    a.dy_sB, a.dy_sH, a.dy_sW = dy4.stride(0), dy4.stride(1), dy4.stride(2)
    a.dw, a.dw_sN, a.accumulate = dw.data_ptr(), dw.stride(0), int(accumulate)
    a.nz, a.nz_inner = 1, 1
    lib = _lib.load()
    nbytes = lib.gdl_conv_wgrad_workspace(C.byref(a))
    ws = torch.empty(max(nbytes, 4) // 4, device=x.device, dtype=torch.float32)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.gdl_conv_wgrad(C.byref(a), _stream()), "gdl_conv_wgrad")
    return dw


# ------------------------------------------------------------------ normalisation
def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, out_dtype: torch.dtype) -> Tensor:
    _need_cuda(x)
    if x.dtype != torch.float32 or x.stride(-1) != 1:
        raise ValueError("layernorm: x must be f32 with unit last stride")
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    y = torch.empty((x2.shape[0], D), device=x.device, dtype=out_dtype)
    check(_lib.load().gdl_layernorm_fwd(_p(x2), x2.stride(0), _p(_f32vec(gamma, D, "gamma")),
                                        _p(_f32vec(beta, D, "beta")), _p(y), dt(y), x2.shape[0], D,
                                        eps, _stream()), "gdl_layernorm_fwd")
    return y.reshape(x.shape)


def _pix(x: Tensor, name: str):
    """NHWC tensor whose pixels are uniformly strided -> (P, C, pixel stride)."""
    x4 = _nhwc4(x, name)
    B, H, W, Cc = x4.shape
    sB, sH, sW = x4.stride(0), x4.stride(1), x4.stride(2)
    if W > 1:
        sP = sW
    elif H > 1:
        sP = sH
    else:
        sP = sB if B > 1 else Cc
    ok = (H == 1 or W == 1 or sH == W * sP) and (B == 1 or H * W == 1 or sB == H * W * sP)
    if not ok:
        raise ValueError(f"{name}: pixels must be uniformly strided, got strides {x4.stride()}")
    return B * H * W, Cc, sP


def bn_stats(x: Tensor, running_mean: Tensor | None = None, running_var: Tensor | None = None,
             momentum: float = 0.1):
    _need_cuda(x)
    P, Cc, sP = _pix(x, "bn_stats x")
    mean = torch.empty(Cc, device=x.device, dtype=torch.float32)
    var = torch.empty_like(mean)
    lib = _lib.load()
    nbytes = lib.gdl_bn_stats_workspace(P, Cc)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    check(lib.gdl_bn_stats(_p(x), dt(x), P, Cc, sP, _p(mean), _p(var), _p(running_mean),
                           _p(running_var), momentum, _p(ws), nbytes, _stream()), "gdl_bn_stats")
    return mean, var


def bn_stats_finalize(partials: Tensor, rows: int, channels: int, pixels: int, running_mean: Tensor | None = None,
                      running_var: Tensor | None = None, momentum: float = 0.1):
    """(mean, biased var) from [rows, 2, C] partial sums written by a producing kernel (conv_gemm(want_stats=True)); updates the
    running buffers like bn_stats."""
    _need_cuda(partials)
    mean = torch.empty(channels, device=partials.device, dtype=torch.float32)
    var = torch.empty_like(mean)
    check(_lib.load().gdl_bn_stats_finalize(_p(partials), rows, channels, pixels, _p(mean), _p(var), _p(running_mean),
                                            _p(running_var), momentum, _stream()), "gdl_bn_stats_finalize")
    return mean, var


def bn_apply(x: Tensor, mean: Tensor, var: Tensor, gamma: Tensor, beta: Tensor, eps: float,
             relu: bool, out: Tensor | None = None) -> Tensor:
    P, Cc, sP = _pix(x, "bn_apply x")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    Po, Co, sPo = _pix(out, "bn_apply out")
    check(_lib.load().gdl_bn_apply(_p(x), _p(out), dt(x), P, Cc, sP, sPo, _p(mean), _p(var),
                                   _p(gamma), _p(beta), eps, int(relu), _stream()), "gdl_bn_apply")
    return out


def syncbn_pack(mean: Tensor, var: Tensor, count: float, out: Tensor) -> None:
    """out[0:C] = count * mean, out[C:2C] = count * (var + mean^2), out[2C] = count: one rank's share of the SyncBatchNorm message
    (out: a contiguous f32 slice of 2 C + 1 elements of the message buffer)."""
    c = mean.numel()
    if out.numel() != 2 * c + 1 or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("syncbn_pack: out must be a contiguous f32 slice of 2 C + 1 elements")
    check(_lib.load().gdl_syncbn_pack(_p(mean), _p(var), float(count), c, _p(out), _stream()), "gdl_syncbn_pack")


def syncbn_unpack(packed: Tensor, channels: int, running_mean: Tensor | None = None, running_var: Tensor | None = None,
                  momentum: float = 0.1):
    """(global mean, global biased var) from the all-reduced message slice [sum n mean | sum n E[x^2] | sum n]; the running
    estimates are updated with the unbiased variance over the global count (which stays on the device)."""
    if packed.numel() != 2 * channels + 1 or packed.dtype != torch.float32 or not packed.is_contiguous():
        raise ValueError("syncbn_unpack: packed must be a contiguous f32 slice of 2 C + 1 elements")
    mean = torch.empty(channels, device=packed.device, dtype=torch.float32)
    var = torch.empty_like(mean)
    check(_lib.load().gdl_syncbn_unpack(_p(packed), channels, _p(mean), _p(var), _p(running_mean), _p(running_var), momentum, _stream()),
          "gdl_syncbn_unpack")
    return mean, var


def bn_bwd_dx_sync(x, dy, mean, var, gamma, beta, eps, relu, dgamma_sum, dbeta_sum, total_count: Tensor, out: Tensor | None = None):
    """bn_bwd_dx with the all-reduced sums and the GLOBAL pixel count read from device memory (``total_count``: one f32 element,
    e.g. entry 2 C of the forward message)."""
    P, Cc, sP = _pix(x, "bn_bwd x")
    _, _, sPd = _pix(dy, "bn_bwd dy")
    dx = out if out is not None else torch.empty(x.shape, device=x.device, dtype=x.dtype)
    _, _, sPx = _pix(dx, "bn_bwd dx")
    if total_count.dtype != torch.float32 or total_count.numel() != 1:
        raise ValueError("bn_bwd_dx_sync: total_count must be one f32 element on the device")
    check(_lib.load().gdl_bn_bwd_dx_sync(_p(x), _p(dy), _p(dx), dt(x), P, Cc, sP, sPd, sPx, _p(mean), _p(var), _p(gamma), _p(beta), eps,
                                         int(relu), _p(dgamma_sum), _p(dbeta_sum), _p(total_count), _stream()), "gdl_bn_bwd_dx_sync")
    return dx


# OFF by default (0): measured at per-GPU batch 4 (profiles/r05d_*), the one-launch kernels make the EAGER step 5-10 % faster
# (fewer launches to issue) but the step replayed from a hipGraph -- the default path at that batch -- 3.8 % SLOWER (447 vs 465
# tiles/s): a workgroup that owns four channels over all pixels reads 8 bytes out of every 128-byte line, and the C / 4
# workgroups sit on different XCDs, so every line crosses the fabric up to sixteen times; the three short many-workgroup
# launches read whole lines.  GDL_BN_SMALL_PIXELS=8192 switches them on (host-bound eager loops without graph capture).
BN_SMALL_MAX_PIXELS = int(os.environ.get("GDL_BN_SMALL_PIXELS", "0"))


def bn_small_ok(x: Tensor) -> bool:
    """Maps the one-launch BatchNorm kernels take: at most BN_SMALL_MAX_PIXELS pixels (each workgroup walks ALL pixels of its
    four channels: beyond a few thousand the many-workgroup kernels win)."""
    if x.dim() < 2 or x.shape[-1] % 4:
        return False
    return 0 < x.numel() // x.shape[-1] <= BN_SMALL_MAX_PIXELS


def bn_small_fits(x: Tensor) -> bool:
    """Shapes the kernels can take at all (tests; the dispatch uses bn_small_ok)."""
    return x.dim() >= 2 and x.shape[-1] % 4 == 0 and x.numel() > 0


def bn_small_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, relu: bool, running_mean: Tensor | None = None,
                 running_var: Tensor | None = None, momentum: float = 0.1):
    """Train-mode BatchNorm(+ReLU) of a small map in ONE launch: (out, mean, biased var); updates the running buffers."""
    _need_cuda(x)
    P, Cc, sP = _pix(x, "bn_small_fwd x")
    out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    _, _, sPo = _pix(out, "bn_small_fwd out")
    mean = torch.empty(Cc, device=x.device, dtype=torch.float32)
    var = torch.empty_like(mean)
    check(_lib.load().gdl_bn_small_fwd(_p(x), _p(out), dt(x), P, Cc, sP, sPo, _p(gamma), _p(beta), eps, int(relu), _p(mean), _p(var),
                                       _p(running_mean), _p(running_var), momentum, _stream()), "gdl_bn_small_fwd")
    return out, mean, var


def bn_small_bwd(x: Tensor, dy: Tensor, mean, var, gamma, beta, eps, relu, out: Tensor | None = None):
    """Backward of bn_small_fwd in ONE launch: (dx, dgamma, dbeta).  ``out`` may be x (dx then replaces the saved conv output)."""
    P, Cc, sP = _pix(x, "bn_small_bwd x")
    _, _, sPd = _pix(dy, "bn_small_bwd dy")
    if dy.dtype != x.dtype:
        raise ValueError("bn_small_bwd: dy dtype must match x")
    dx = out if out is not None else torch.empty(x.shape, device=x.device, dtype=x.dtype)
    _, _, sPx = _pix(dx, "bn_small_bwd dx")
    dgamma = torch.empty(Cc, device=x.device, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    check(_lib.load().gdl_bn_small_bwd(_p(x), _p(dy), _p(dx), dt(x), P, Cc, sP, sPd, sPx, _p(mean), _p(var), _p(gamma), _p(beta), eps,
                                       int(relu), _p(dgamma), _p(dbeta), _stream()), "gdl_bn_small_bwd")
    return dx, dgamma, dbeta


def bn_bwd_reduce(x, dy, mean, var, gamma, beta, eps, relu):
    P, Cc, sP = _pix(x, "bn_bwd x")
    _, _, sPd = _pix(dy, "bn_bwd dy")
    if dy.dtype != x.dtype:
        raise ValueError("bn_bwd: dy dtype must match x")
    dgamma = torch.empty(Cc, device=x.device, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    lib = _lib.load()
    nbytes = lib.gdl_bn_stats_workspace(P, Cc)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    check(lib.gdl_bn_bwd_reduce(_p(x), _p(dy), dt(x), P, Cc, sP, sPd, _p(mean), _p(var), _p(gamma),
                                _p(beta), eps, int(relu), _p(dgamma), _p(dbeta), _p(ws), nbytes,
                                _stream()), "gdl_bn_bwd_reduce")
    return dgamma, dbeta


def bn_bwd_dx(x, dy, mean, var, gamma, beta, eps, relu, dgamma_sum, dbeta_sum, p_total,
              out: Tensor | None = None):
    P, Cc, sP = _pix(x, "bn_bwd x")
    _, _, sPd = _pix(dy, "bn_bwd dy")
    dx = out if out is not None else torch.empty(x.shape, device=x.device, dtype=x.dtype)
    _, _, sPx = _pix(dx, "bn_bwd dx")
    check(_lib.load().gdl_bn_bwd_dx(_p(x), _p(dy), _p(dx), dt(x), P, Cc, sP, sPd, sPx, _p(mean),
                                    _p(var), _p(gamma), _p(beta), eps, int(relu), _p(dgamma_sum),
                                    _p(dbeta_sum), p_total, _stream()), "gdl_bn_bwd_dx")
    return dx


# ------------------------------------------------------------------ resampling
def bilinear(x: Tensor, size: tuple[int, int], out: Tensor | None = None,
             out_dtype: torch.dtype | None = None, accumulate: bool = False) -> Tensor:
    _need_cuda(x)
    x4 = _nhwc4(x, "bilinear x")
    B, Hi, Wi, Cc = x4.shape
    Ho, Wo = size
    if out is None:
        if accumulate:
            raise ValueError("bilinear: accumulate needs out")
        out = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=out_dtype or x.dtype)
    o4 = _nhwc4(out, "bilinear out")
    if tuple(o4.shape) != (B, Ho, Wo, Cc):
        raise ValueError(f"bilinear: out shape {tuple(o4.shape)} != {(B, Ho, Wo, Cc)}")
    check(_lib.load().gdl_bilinear_fwd(_p(x4), dt(x4), B, Hi, Wi, Cc, x4.stride(0), x4.stride(1),
                                       x4.stride(2), _p(o4), dt(o4), Ho, Wo, o4.stride(0),
                                       o4.stride(1), o4.stride(2), int(accumulate), _stream()),
          "gdl_bilinear_fwd")
    return out


def bilinear_add(base: Tensor, x: Tensor) -> Tensor:
    """base + bilinear(x -> base's size) as a new tensor, in one pass where the fused kernel applies (gdl_bilinear_fwd_add;
    otherwise copy + accumulate inside the library): UperNet's top-down add (upernet.py:127-135)."""
    _need_cuda(base, x)
    b4, x4 = _nhwc4(base, "bilinear_add base"), _nhwc4(x, "bilinear_add x")
    B, Ho, Wo, Cc = b4.shape
    if x4.shape[0] != B or x4.shape[3] != Cc or x4.dtype != b4.dtype:
        raise ValueError(f"bilinear_add: {tuple(x4.shape)} {x4.dtype} does not resize onto {tuple(b4.shape)} {b4.dtype}")
    b4 = b4 if b4.is_contiguous() else b4.contiguous()
    out = torch.empty((B, Ho, Wo, Cc), device=base.device, dtype=base.dtype)
    check(_lib.load().gdl_bilinear_fwd_add(_p(x4), dt(x4), B, x4.shape[1], x4.shape[2], Cc, x4.stride(0), x4.stride(1), x4.stride(2),
                                           _p(b4), _p(out), dt(out), Ho, Wo, out.stride(0), out.stride(1), out.stride(2), _stream()),
          "gdl_bilinear_fwd_add")
    return out


GATHER_TWO_PASS = True    # A/B switch (tools / tests): separable two-pass gather for large resize factors


def resize_conv3x3_bwd_gather(dy: Tensor, in_size: tuple[int, int]) -> Tensor:
    """The nine low-resolution maps G_t = resize^T(shift_t^T(dy)) of conv3x3(pad 1)(bilinear resize(x)) as one dense
    [B, Hi, Wi, 9 * N] tensor, tap block 8 - t (see gdl_resize_conv3x3_bwd_gather)."""
    _need_cuda(dy)
    d4 = _nhwc4(dy, "resize_conv3x3_bwd_gather dy")
    if not d4.is_contiguous():
        raise ValueError("resize_conv3x3_bwd_gather: dy must be dense NHWC")
    B, Ho, Wo, N = d4.shape
    Hi, Wi = in_size
    g = torch.empty((B, Hi, Wi, 9 * N), device=dy.device, dtype=dy.dtype)
    lib = _lib.load()
    one_pass = lib.gdl_resize_conv3x3_bwd_gather_one_pass(dt(d4), B, Ho, Wo, N, Hi, Wi)     # matrix-core form (bf16, x2 / x4)
    if GATHER_TWO_PASS and Ho >= 2 * Hi and not one_pass:      # upsampling factors >= 2: the single VALU pass is multiply-add bound
        nbytes = lib.gdl_resize_conv3x3_bwd_gather_workspace(dt(d4), B, Wo, N, Hi)
        ws = torch.empty(nbytes // d4.element_size(), device=dy.device, dtype=dy.dtype)
        check(lib.gdl_resize_conv3x3_bwd_gather2(_p(d4), dt(d4), B, Ho, Wo, N, _p(g), Hi, Wi, _p(ws), nbytes, _stream()),
              "gdl_resize_conv3x3_bwd_gather2")
    else:
        check(lib.gdl_resize_conv3x3_bwd_gather(_p(d4), dt(d4), B, Ho, Wo, N, _p(g), Hi, Wi, _stream()),
              "gdl_resize_conv3x3_bwd_gather")
    return g


def resize_conv3x3_bwd_gather_bn_ok(dz: Tensor, in_size: tuple[int, int]) -> bool:
    """Shapes the BatchNorm-backward-fused gather takes (= the one-pass matrix-core gather: bf16, N % 64 == 0, factor 2 / 4)."""
    if dz.dim() != 4 or dz.dtype != torch.bfloat16 or not dz.is_contiguous():
        return False
    B, Ho, Wo, N = dz.shape
    return bool(_lib.load().gdl_resize_conv3x3_bwd_gather_one_pass(dt(dz), B, Ho, Wo, N, in_size[0], in_size[1]))


def resize_conv3x3_bwd_gather_bn(dz: Tensor, y: Tensor, in_size: tuple[int, int], mean, var, gamma, beta, eps, relu, dgamma_sum,
                                 dbeta_sum, p_total) -> Tensor:
    """resize_conv3x3_bwd_gather(bn_bwd_dx(y, dz, ...)) in one kernel: the BatchNorm(+ReLU) backward is applied to dz while it is
    staged, its result is never written (gdl_resize_conv3x3_bwd_gather_bn).  dz, y dense NHWC bf16 of one shape."""
    _need_cuda(dz, y)
    if dz.shape != y.shape or dz.dtype != y.dtype or not (dz.is_contiguous() and y.is_contiguous()):
        raise ValueError("resize_conv3x3_bwd_gather_bn: dz and y must be dense NHWC tensors of one shape and dtype")
    B, Ho, Wo, N = dz.shape
    Hi, Wi = in_size
    g = torch.empty((B, Hi, Wi, 9 * N), device=dz.device, dtype=dz.dtype)
    coef = torch.empty(4 * N, device=dz.device, dtype=torch.float32)
    check(_lib.load().gdl_resize_conv3x3_bwd_gather_bn(_p(dz), _p(y), dt(dz), B, Ho, Wo, N, _p(g), Hi, Wi, _p(mean), _p(var), _p(gamma),
                                                       _p(beta), eps, int(relu), _p(dgamma_sum), _p(dbeta_sum), p_total, _p(coef),
                                                       _stream()), "gdl_resize_conv3x3_bwd_gather_bn")
    return g


def resize_conv3x3_bwd(x_lo: Tensor, dy: Tensor | None, w_dgrad: Tensor | None, want_dw: bool = True, g: Tensor | None = None):
    """(dx_lo, dw) of y = conv3x3(pad 1)(bilinear resize(x_lo -> dy's size)) from dy, as GEMMs over the LOW-resolution
    pixels.  ``w_dgrad`` [C, 9 * N] (gdl_pack_dgrad operand; None = no data gradient); dw [N, 9 * C] f32.  ``g``: the nine
    gathered maps when the caller already has them (resize_conv3x3_bwd_gather_bn), dy is then unused."""
    x4 = _nhwc4(x_lo, "resize_conv3x3_bwd x")
    B, Hi, Wi, Cc = x4.shape
    if g is None:
        g = resize_conv3x3_bwd_gather(dy, (Hi, Wi))
    N = g.shape[-1] // 9
    dx = conv_gemm(g, w_dgrad) if w_dgrad is not None else None
    dw = None
    if want_dw:
        # one 1x1 weight gradient with the nine maps as 9 N "output channels": [(8 - t, n), c] -> [n, (t, c)]
        dwp = conv_wgrad(x4, g, R=1, S=1)
        dw = dwp.view(9, N, Cc).flip(0).permute(1, 0, 2).reshape(N, 9 * Cc)
    return dx, dw


def resize_conv3x3_fwd_sum(zs: list[Tensor], size: tuple[int, int], addvec: Tensor | None = None, relu: bool = False) -> Tensor:
    """sum_k sum_t shift_t(bilinear(zs[k][..., t*N:(t+1)*N] -> size)) (+ addvec, ReLU): the pixel side of
    conv3x3(pad 1)(bilinear resize(x)) once the nine tap products z = [W_0 x, ..., W_8 x] exist at low resolution
    (gdl_resize_conv3x3_fwd_sum).  zs: 1..3 dense [B, h_k, w_k, 9 N] maps of one dtype.  Sources whose size divides ``size`` by
    2 / 4 / 8 go through the cell kernels (matrix cores in bf16) in ONE pass; any other ratio (DOFA-large's 36 -> 292) is added
    by the plain gather kernel, one pass per such source (gdl_resize_conv3x3_fwd_sum_any)."""
    _need_cuda(*zs)
    if not 1 <= len(zs) <= 3:
        raise ValueError("resize_conv3x3_fwd_sum: 1..3 sources")
    z4 = [_nhwc4(z, "resize_conv3x3_fwd_sum source") for z in zs]
    B, N9 = z4[0].shape[0], z4[0].shape[3]
    if N9 % 9:
        raise ValueError("resize_conv3x3_fwd_sum: sources must have 9 * N channels")
    N = N9 // 9
    for z in z4:
        if z.shape[0] != B or z.shape[3] != N9 or z.dtype != z4[0].dtype or not z.is_contiguous():
            raise ValueError("resize_conv3x3_fwd_sum: sources must be contiguous NHWC maps with equal batch, channels and dtype")
    out = torch.empty((B, size[0], size[1], N), device=zs[0].device, dtype=zs[0].dtype)
    cell = [z for z in z4 if resize_conv3x3_fwd_ok((z.shape[1], z.shape[2]), size, B)]
    other = [z for z in z4 if not resize_conv3x3_fwd_ok((z.shape[1], z.shape[2]), size, B)]
    lib = _lib.load()
    av = _f32vec(addvec, N, "addvec")
    if cell:
        n = len(cell)
        ptrs = (C.c_void_p * 3)(*([z.data_ptr() for z in cell] + [None] * (3 - n)))
        hs = (C.c_int * 3)(*([z.shape[1] for z in cell] + [1] * (3 - n)))
        ws = (C.c_int * 3)(*([z.shape[2] for z in cell] + [1] * (3 - n)))
        # (addvec and the ReLU belong to the LAST pass over `out`)
        check(lib.gdl_resize_conv3x3_fwd_sum(ptrs, hs, ws, n, dt(cell[0]), B, N, _p(out), size[0], size[1],
                                             _p(None if other else av), int(relu and not other), _stream()),
              "gdl_resize_conv3x3_fwd_sum")
    for i, z in enumerate(other):
        last = i == len(other) - 1
        check(lib.gdl_resize_conv3x3_fwd_sum_any(_p(z), z.shape[1], z.shape[2], dt(z), B, N, _p(out), size[0], size[1],
                                                 int(bool(cell) or i > 0), _p(av if last else None), int(relu and last), _stream()),
              "gdl_resize_conv3x3_fwd_sum_any")
    return out


def resize_conv3x3_fwd_sum_bn(zs: list[Tensor], size: tuple[int, int], addvec: Tensor | None = None,
                              running_mean: Tensor | None = None, running_var: Tensor | None = None, momentum: float = 0.1):
    """resize_conv3x3_fwd_sum + the train-mode BatchNorm statistics of its result in the same pass
    (gdl_resize_conv3x3_fwd_sum_bn): returns (out, mean, biased var); updates the running buffers when given.  bf16, N % 64 == 0."""
    _need_cuda(*zs)
    z4 = [_nhwc4(z, "resize_conv3x3_fwd_sum_bn source") for z in zs]
    B, N9 = z4[0].shape[0], z4[0].shape[3]
    N = N9 // 9
    for z in z4:
        if z.shape[0] != B or z.shape[3] != N9 or z.dtype != z4[0].dtype or not z.is_contiguous():
            raise ValueError("resize_conv3x3_fwd_sum_bn: sources must be contiguous NHWC maps with equal batch, channels and dtype")
    out = torch.empty((B, size[0], size[1], N), device=zs[0].device, dtype=zs[0].dtype)
    mean = torch.empty(N, device=out.device, dtype=torch.float32)
    var = torch.empty_like(mean)
    lib = _lib.load()
    rows = lib.gdl_resize_conv3x3_fwd_sum_bn_rows(B, size[0], size[1])
    wsp = torch.empty(rows * 2 * N, device=out.device, dtype=torch.float32)
    n = len(z4)
    ptrs = (C.c_void_p * 3)(*([z.data_ptr() for z in z4] + [None] * (3 - n)))
    hs = (C.c_int * 3)(*([z.shape[1] for z in z4] + [1] * (3 - n)))
    ws = (C.c_int * 3)(*([z.shape[2] for z in z4] + [1] * (3 - n)))
    check(lib.gdl_resize_conv3x3_fwd_sum_bn(ptrs, hs, ws, n, dt(z4[0]), B, N, _p(out), size[0], size[1],
                                            _p(_f32vec(addvec, N, "addvec")), _p(wsp), wsp.numel() * 4, _p(mean), _p(var),
                                            _p(running_mean), _p(running_var), momentum, _stream()),
          "gdl_resize_conv3x3_fwd_sum_bn")
    return out, mean, var


def resize_conv3x3_fwd_bn_ok(z_dtype: torch.dtype, N: int, addvec: Tensor | None = None) -> bool:
    """The statistics variant runs on the matrix-core kernel only: bf16, N % 64 == 0 and a per-channel addend (bias) the
    kernel can fetch with 16-byte loads (a bias that is a view at an odd offset falls back to the separate statistics pass)."""
    if addvec is not None and addvec.data_ptr() % 16:
        return False
    return z_dtype == torch.bfloat16 and N % 64 == 0


def resize_conv3x3_fwd_ok(lo: tuple[int, int], size: tuple[int, int], B: int) -> bool:
    """Shapes gdl_resize_conv3x3_fwd_sum takes: one integer factor of 2, 4 or 8 in both directions, B * rows in one grid dim."""
    f = size[0] // max(lo[0], 1)
    return f in (2, 4, 8) and lo[0] * f == size[0] and lo[1] * f == size[1] and B * size[0] <= 65535


MAX_ANY_RESIZE = 10.0     # the backward gather's widest window instantiation (2 * ceil(factor) + 4 <= 24 columns)


def resize_conv3x3_any_ok(lo: tuple[int, int], size: tuple[int, int], B: int) -> bool:
    """Upsampling ratios the low-resolution forms of conv3x3(resize(x)) take at all: any real factor in (1, 10] per direction
    (forward: resize_conv3x3_fwd_sum's plain gather pass; backward: the gather kernels' window instantiations)."""
    fy, fx = size[0] / max(lo[0], 1), size[1] / max(lo[1], 1)
    return 1.0 < fy <= MAX_ANY_RESIZE and 1.0 < fx <= MAX_ANY_RESIZE and B * size[0] <= 65535


def copy_cast(x: Tensor, out: Tensor | None = None, out_dtype: torch.dtype | None = None) -> Tensor:
    """Strided NHWC copy with dtype conversion (f32 / bf16): dense copy of a channel slice, compute-dtype cast."""
    _need_cuda(x)
    x4 = _nhwc4(x, "copy_cast x")
    B, H, W, Cc = x4.shape
    if out is None:
        out = torch.empty(x4.shape, device=x.device, dtype=out_dtype or x.dtype)
    o4 = _nhwc4(out, "copy_cast out")
    if tuple(o4.shape) != (B, H, W, Cc):
        raise ValueError(f"copy_cast: out shape {tuple(o4.shape)} != {(B, H, W, Cc)}")
    if B * H > 65535 or Cc % 4:        # very tall batches / odd channel counts: the identity resample handles any layout
        return bilinear(x, (H, W), out=out)      # (same HBM-bound copy through the flat-index kernel: no slow path to warn about)
    check(_lib.load().gdl_copy_cast(_p(x4), dt(x4), B, H, W, Cc, x4.stride(0), x4.stride(1), x4.stride(2), _p(o4), dt(o4),
                                    o4.stride(0), o4.stride(1), o4.stride(2), _stream()), "gdl_copy_cast")
    return out


def bilinear_sum(xs: list[Tensor], size: tuple[int, int]) -> Tensor:
    """sum_k bilinear(xs[k] -> size) for 1..3 dense NHWC maps with equal batch / channels / dtype, one output write."""
    _need_cuda(*xs)
    if not 1 <= len(xs) <= 3:
        raise ValueError("bilinear_sum: 1..3 sources")
    x4s = [_nhwc4(x, "bilinear_sum source") for x in xs]
    B, _, _, Cc = x4s[0].shape
    for x in x4s:
        if x.shape[0] != B or x.shape[3] != Cc or x.dtype != x4s[0].dtype or not x.is_contiguous():
            raise ValueError("bilinear_sum: sources must be contiguous NHWC maps with equal batch, channels and dtype")
    out = torch.empty((B, size[0], size[1], Cc), device=xs[0].device, dtype=xs[0].dtype)
    n = len(x4s)
    ptrs = (C.c_void_p * 3)(*([x.data_ptr() for x in x4s] + [None] * (3 - n)))
    hs = (C.c_int * 3)(*([x.shape[1] for x in x4s] + [1] * (3 - n)))
    ws = (C.c_int * 3)(*([x.shape[2] for x in x4s] + [1] * (3 - n)))
    check(_lib.load().gdl_bilinear_sum_fwd(ptrs, hs, ws, n, dt(x4s[0]), B, Cc, _p(out), size[0], size[1], _stream()),
          "gdl_bilinear_sum_fwd")
    return out


def bilinear_bwd(dout: Tensor, in_size: tuple[int, int], din: Tensor | None = None,
                 din_dtype: torch.dtype | None = None, accumulate: bool = False) -> Tensor:
    d4 = _nhwc4(dout, "bilinear_bwd dout")
    B, Ho, Wo, Cc = d4.shape
    Hi, Wi = in_size
    if din is None:
        din = torch.empty((B, Hi, Wi, Cc), device=dout.device, dtype=din_dtype or dout.dtype)
    i4 = _nhwc4(din, "bilinear_bwd din")
    check(_lib.load().gdl_bilinear_bwd(_p(d4), dt(d4), B, Ho, Wo, Cc, d4.stride(0), d4.stride(1),
                                       d4.stride(2), _p(i4), dt(i4), Hi, Wi, i4.stride(0),
                                       i4.stride(1), i4.stride(2), int(accumulate), _stream()),
          "gdl_bilinear_bwd")
    return din


def adaptive_avgpool(x: Tensor, s: int, out_dtype: torch.dtype | None = None) -> Tensor:
    x4 = _nhwc4(x, "avgpool x")
    B, Hi, Wi, Cc = x4.shape
    out = torch.empty((B, s, s, Cc), device=x.device, dtype=out_dtype or x.dtype)
    check(_lib.load().gdl_adaptive_avgpool_fwd(_p(x4), dt(x4), B, Hi, Wi, Cc, x4.stride(0),
                                               x4.stride(1), x4.stride(2), _p(out), dt(out), s,
                                               _stream()), "gdl_adaptive_avgpool_fwd")
    return out


def adaptive_avgpool_bwd(dout: Tensor, in_size: tuple[int, int], din: Tensor | None = None,
                         accumulate: bool = False) -> Tensor:
    if not dout.is_contiguous():
        raise ValueError("adaptive_avgpool_bwd: dout must be contiguous NHWC")
    B, s, _, Cc = dout.shape
    Hi, Wi = in_size
    if din is None:
        din = torch.empty((B, Hi, Wi, Cc), device=dout.device, dtype=dout.dtype)
    i4 = _nhwc4(din, "avgpool_bwd din")
    check(_lib.load().gdl_adaptive_avgpool_bwd(_p(dout), dt(dout), B, s, Cc, _p(i4), dt(i4), Hi, Wi,
                                               i4.stride(0), i4.stride(1), i4.stride(2),
                                               int(accumulate), _stream()),
          "gdl_adaptive_avgpool_bwd")
    return din


# ------------------------------------------------------------------ fused bilinear x4 upsample -> 3x3 conv
def pad_nhwc(x: Tensor, pad_h: int, pad_w: int, zero: bool = False) -> Tensor:
    """NHWC border padding: replicate (default) or zeros."""
    _need_cuda(x)
    x4 = _nhwc4(x, "pad_nhwc x")
    B, H, W, Cc = x4.shape
    out = torch.empty((B, H + 2 * pad_h, W + 2 * pad_w, Cc), device=x.device, dtype=x.dtype)
    check(_lib.load().gdl_pad_nhwc(_p(x4), dt(x4), B, H, W, Cc, x4.stride(0), x4.stride(1), x4.stride(2), _p(out),
                                   pad_h, pad_w, int(zero), _stream()), "gdl_pad_nhwc")
    return out


def subpix4_weights(w32: Tensor, Cc: int, out_dtype: torch.dtype) -> dict:
    """Phase weights of (bilinear x4 upsample -> 3x3 conv) from w32 [N, 9*C] f32 (K order dy,dx,c); csrc/subpixel.hip."""
    _need_cuda(w32)
    N = w32.shape[0]
    if w32.dtype != torch.float32 or not w32.is_contiguous() or w32.shape[1] != 9 * Cc:
        raise ValueError("subpix4_weights: contiguous f32 [N, 9*C] expected")
    mk = lambda taps: torch.empty((4, N, taps * Cc), device=w32.device, dtype=out_dtype)  # noqa: E731
    out = {"g22": mk(4), "g23": mk(6), "g32": mk(6), "g33": mk(9), "lines": mk(3)}
    check(_lib.load().gdl_subpix4_weights(_p(w32), N, Cc, dt(out["g22"]), _p(out["g22"]), _p(out["g23"]), _p(out["g32"]),
                                          _p(out["g33"]), _p(out["lines"]), _stream()), "gdl_subpix4_weights")
    return out


def up4_conv3x3(x: Tensor, wsets: dict, *, bias: Tensor | None = None, scale: Tensor | None = None,
                shift: Tensor | None = None, act: int = ACT_NONE) -> Tensor:
    """conv3x3(pad 1)(bilinear_x4(x)) on NHWC x [B,H,W,C] WITHOUT materialising the upsampled map: 16 phase
    convolutions of the replicate-padded low-res map (four batched launches, 6.25 low-res taps on average instead of 9
    high-res ones) written into the strided phase positions of the output, then the four outermost output lines --
    the only ones the convolution's zero padding touches -- recomputed exactly by 1x3 line convolutions.
    Reference: multilevel_neck.py:157-158 with scale 4 (+ models/utils.py:131-137)."""
    _need_cuda(x)
    x4 = _nhwc4(x, "up4_conv3x3 x")
    B, H, W, Cc = x4.shape
    N = wsets["g22"].shape[1]
    es = x4.element_size()
    xp = pad_nhwc(x4, 1, 1)                                   # replicate: the bilinear index clamping
    Hp, Wp = H + 2, W + 2
    y = torch.empty((B, 4 * H, 4 * W, N), device=x.device, dtype=x.dtype)
    f32 = lambda t, name: None if t is None else _f32vec(t, N, name).data_ptr()  # noqa: E731
    for gy, R in ((0, 2), (1, 3)):
        for gx, S in ((0, 2), (1, 3)):
            wg = wsets[f"g{R}{S}"]
            py0, dpy = (0, 3) if R == 2 else (1, 1)
            px0, dpx = (0, 3) if S == 2 else (1, 1)
            a = ConvArgs()
            a.inp, a.dtype = xp.data_ptr(), dt(xp)
            a.B, a.H, a.W, a.C = B, H + R - 1, W + S - 1, Cc
            a.in_sB, a.in_sH, a.in_sW = Hp * Wp * Cc, Wp * Cc, Cc
            a.Ho, a.Wo, a.R, a.S, a.stride, a.pad = H, W, R, S, 1, 0
            a.w, a.w_sN, a.N = wg.data_ptr(), R * S * Cc, N
            a.out = y.data_ptr() + (py0 * 4 * W * N + px0 * N) * es
            a.out_dtype = dt(y)
            a.out_sB, a.out_sH, a.out_sW = 16 * H * W * N, 16 * W * N, 4 * N
            a.alpha, a.act = 1.0, act
            a.bias, a.scale, a.shift = f32(bias, "bias"), f32(scale, "scale"), f32(shift, "shift")
            a.nz, a.nz_inner = 4, 2                           # z = (row phase index, column phase index)
            a.in_sZ0, a.in_sZ1 = (Wp * Cc if R == 2 else 0), (Cc if S == 2 else 0)
            a.w_sZ0, a.w_sZ1 = 2 * N * R * S * Cc, N * R * S * Cc
            a.out_sZ0, a.out_sZ1 = dpy * 4 * W * N, dpx * N
            batched_gemm_raw(a)
    lines = wsets["lines"]
    for side, (ln, view) in enumerate(zip(_up4_lines(x4), _up4_line_views(y))):
        conv_gemm(ln, lines[side], R=1, S=3, out=view, bias=bias, scale=scale, shift=shift, act=act)
    return y


def _up4_lines(x4: Tensor):
    """The four zero-extended border lines of the x4-upsampled map (top, bottom, left, right), each [B,1,4L+2,C]."""
    B, H, W, Cc = x4.shape
    out = []
    for src in (x4[:, 0:1], x4[:, H - 1:H]):
        out.append(pad_nhwc(bilinear(src, (1, 4 * W)), 0, 1, zero=True))
    for src in (x4[:, :, 0:1], x4[:, :, W - 1:W]):
        out.append(pad_nhwc(bilinear(src, (4 * H, 1)).view(B, 1, 4 * H, Cc), 0, 1, zero=True))
    return out


def _up4_line_views(y: Tensor):
    """The matching output lines of y [B,4H,4W,N] as [B,1,L,N] views."""
    Hh, Ww = y.shape[1], y.shape[2]
    return [y[:, 0:1], y[:, Hh - 1:Hh], y[:, :, 0, :].unsqueeze(1), y[:, :, Ww - 1, :].unsqueeze(1)]


# ------------------------------------------------------------------ attention (unfused + fused)
def _attn_check(q: Tensor, k: Tensor, v: Tensor, num_heads: int):
    _need_cuda(q, k, v)
    if q.dim() != 3 or k.dim() != 3 or v.shape != k.shape or q.shape[0] != k.shape[0] or q.shape[2] != k.shape[2]:
        raise ValueError("attention: expected q [B,Nq,D], k/v [B,Nkv,D]")
    if not (q.dtype == k.dtype == v.dtype) or q.stride(2) != 1 or k.stride(2) != 1 or v.stride(2) != 1:
        raise ValueError("attention: q/k/v must share a dtype and have unit channel stride")
    B, Nq, D = q.shape
    return B, Nq, k.shape[1], D, D // num_heads


def _v_transposed(v: Tensor, num_heads: int, hd: int, npad: int) -> Tensor:
    B, Nkv, _ = v.shape
    vt = torch.empty((B, num_heads, hd, npad), device=v.device, dtype=v.dtype)
    check(_lib.load().gdl_v_transpose(_p(v), dt(v), B, Nkv, num_heads, hd, v.stride(0), v.stride(1), _p(vt),
                                      npad, _stream()), "gdl_v_transpose")
    return vt


def attention_unfused(q: Tensor, k: Tensor, v: Tensor, num_heads: int) -> Tensor:
    """softmax(q k^T / sqrt(hd)) v with materialised scores: the exact-f32 parity path (and the
    small / odd-head-dim cases).  q [B,Nq,D], k,v [B,Nkv,D] (strided views allowed, e.g. slices of
    a packed qkv or kv tensor).  Returns [B,Nq,D]."""
    B, Nq, Nkv, D, hd = _attn_check(q, k, v, num_heads)
    es = 4 if q.dtype == torch.float32 else 2
    al = 16 // es
    if hd % al != 0:
        raise ValueError(f"attention: head_dim {hd} must be a multiple of {al} for {q.dtype}")
    npad = (Nkv + 63) // 64 * 64
    lib = _lib.load()
    vt = _v_transposed(v, num_heads, hd, npad)
    scores = torch.empty((B, num_heads, Nq, npad), device=q.device, dtype=q.dtype)
    a = ConvArgs()
    a.inp, a.dtype = q.data_ptr(), dt(q)
    a.B, a.H, a.W, a.C = 1, 1, Nq, hd
    a.in_sB, a.in_sH, a.in_sW = Nq * q.stride(1), Nq * q.stride(1), q.stride(1)
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad = 1, Nq, 1, 1, 1, 0
    a.w, a.w_sN, a.N = k.data_ptr(), k.stride(1), Nkv
    a.out, a.out_dtype = scores.data_ptr(), dt(scores)
    a.out_sB, a.out_sH, a.out_sW = Nq * npad, Nq * npad, npad
    a.alpha, a.act = float(hd) ** -0.5, ACT_NONE
    a.nz, a.nz_inner = B * num_heads, num_heads
    a.in_sZ0, a.in_sZ1 = q.stride(0), hd
    a.w_sZ0, a.w_sZ1 = k.stride(0), hd
    a.out_sZ0, a.out_sZ1 = num_heads * Nq * npad, Nq * npad
    batched_gemm_raw(a)
    check(lib.gdl_softmax_rows(_p(scores), _p(scores), dt(scores), B * num_heads * Nq, Nkv, npad,
                               _stream()), "gdl_softmax_rows")
    out = torch.empty((B, Nq, D), device=q.device, dtype=q.dtype)
    a = ConvArgs()
    a.inp, a.dtype = scores.data_ptr(), dt(scores)
    a.B, a.H, a.W, a.C = 1, 1, Nq, npad
    a.in_sB, a.in_sH, a.in_sW = Nq * npad, Nq * npad, npad
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad = 1, Nq, 1, 1, 1, 0
    a.w, a.w_sN, a.N = vt.data_ptr(), npad, hd
    a.out, a.out_dtype = out.data_ptr(), dt(out)
    a.out_sB, a.out_sH, a.out_sW = Nq * D, Nq * D, D
    a.alpha, a.act = 1.0, ACT_NONE
    a.nz, a.nz_inner = B * num_heads, num_heads
    a.in_sZ0, a.in_sZ1 = num_heads * Nq * npad, Nq * npad
    a.w_sZ0, a.w_sZ1 = num_heads * hd * npad, hd * npad
    a.out_sZ0, a.out_sZ1 = Nq * D, hd
    batched_gemm_raw(a)
    return out


def attention_flash(q: Tensor, k: Tensor, v: Tensor, num_heads: int, return_lse: bool = False):
    """Fused flash attention forward (bf16, head_dim 64); same argument convention.  V is read row-major (transposed
    inside the LDS).  ``return_lse``: also return the log-sum-exp of the scaled scores [B,H,Nq] f32, which the fused
    backward recomputes the probabilities from."""
    B, Nq, Nkv, D, hd = _attn_check(q, k, v, num_heads)
    if q.dtype != torch.bfloat16 or hd != 64:
        raise ValueError("attention_flash: needs bf16 and head_dim 64")
    out = torch.empty((B, Nq, D), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, num_heads, Nq), device=q.device, dtype=torch.float32) if return_lse else None
    check(_lib.load().gdl_flash_attn_fwd2(_p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1),
                                          _p(v), v.stride(0), v.stride(1), _p(out), out.stride(0), out.stride(1),
                                          _p(lse), B, num_heads, Nq, Nkv, float(hd) ** -0.5, _stream()),
          "gdl_flash_attn_fwd2")
    return (out, lse) if return_lse else out


def attention_flash_v1(q: Tensor, k: Tensor, v: Tensor, num_heads: int) -> Tensor:
    """First-generation kernel (separate V^T pass); kept for A/B measurements."""
    B, Nq, Nkv, D, hd = _attn_check(q, k, v, num_heads)
    npad = (Nkv + 63) // 64 * 64
    vt = _v_transposed(v, num_heads, hd, npad)
    out = torch.empty((B, Nq, D), device=q.device, dtype=q.dtype)
    check(_lib.load().gdl_flash_attn_fwd(_p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1),
                                         _p(vt), _p(out), B, num_heads, Nq, Nkv, npad, float(hd) ** -0.5,
                                         _stream()), "gdl_flash_attn_fwd")
    return out


def flash_ok(q: Tensor, num_heads: int) -> bool:
    return q.dtype == torch.bfloat16 and q.shape[2] // num_heads == 64


def attention(q: Tensor, k: Tensor, v: Tensor, num_heads: int, return_lse: bool = False):
    """Dispatch: flash kernel for bf16 / head_dim 64, materialised-score path otherwise (lse = None there)."""
    if flash_ok(q, num_heads):
        return attention_flash(q, k, v, num_heads, return_lse)
    out = attention_unfused(q, k, v, num_heads)
    return (out, None) if return_lse else out


def attention_flash_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse: Tensor, num_heads: int,
                        dq: Tensor, dk: Tensor, dv: Tensor) -> None:
    """Fused attention backward (bf16, head_dim 64): dq / dk / dv written in place (strided slices allowed)."""
    B, Nq, Nkv, D, hd = _attn_check(q, k, v, num_heads)
    for t, n in ((o, Nq), (do, Nq), (dq, Nq), (dk, Nkv), (dv, Nkv)):
        if t.shape != (B, n, D) or t.dtype != torch.bfloat16 or t.stride(2) != 1:
            raise ValueError("attention_flash_bwd: tensor shape / dtype / stride mismatch")
    if lse.shape != (B, num_heads, Nq) or lse.dtype != torch.float32 or not lse.is_contiguous():
        raise ValueError("attention_flash_bwd: lse must be contiguous f32 [B,H,Nq]")
    dvec = torch.empty_like(lse)
    nbytes = _lib.load().gdl_flash_attn_bwd_workspace(B, num_heads, Nq, Nkv)
    ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32) if nbytes else None
    check(_lib.load().gdl_flash_attn_bwd(
        _p(q), q.stride(0), q.stride(1), _p(k), k.stride(0), k.stride(1), _p(v), v.stride(0), v.stride(1),
        _p(o), o.stride(0), o.stride(1), _p(do), do.stride(0), do.stride(1), _p(lse), _p(dvec),
        _p(dq), dq.stride(0), dq.stride(1), _p(dk), dk.stride(0), dk.stride(1), _p(dv), dv.stride(0), dv.stride(1),
        B, num_heads, Nq, Nkv, float(hd) ** -0.5, _p(ws), nbytes, _stream()), "gdl_flash_attn_bwd")


def split_qkv(qkv: Tensor):
    """[B,N,3D] packed timm qkv -> three strided views [B,N,D] (no copy)."""
    D = qkv.shape[-1] // 3
    return qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]


def dwconv3x3(x: Tensor, w9: Tensor, bias: Tensor, gelu: bool, out_dtype: torch.dtype | None = None) -> Tensor:
    """Depthwise 3x3 (pad 1) + bias (+GELU) on a contiguous NHWC tensor; w9 is [9, C] f32."""
    _need_cuda(x)
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("dwconv3x3: contiguous NHWC input expected")
    B, H, W, Cc = x.shape
    if w9.shape != (9, Cc) or w9.dtype != torch.float32 or not w9.is_contiguous():
        raise ValueError("dwconv3x3: w9 must be contiguous f32 [9, C]")
    out = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    check(_lib.load().gdl_dwconv3x3(_p(x), dt(x), B, H, W, Cc, _p(w9), _p(_f32vec(bias, Cc, "bias")), int(gelu),
                                    _p(out), dt(out), _stream()), "gdl_dwconv3x3")
    return out


# ------------------------------------------------------------------ transformer-block backward
def _colreduce_ws(rows: int, Cc: int, planes: int, device) -> tuple[Tensor, int]:
    nbytes = _lib.load().gdl_colreduce_workspace(rows, Cc, planes)
    return torch.empty(max(nbytes, 4) // 4, device=device, dtype=torch.float32), nbytes


def layernorm_bwd(x: Tensor, dy: Tensor, gamma: Tensor, eps: float, dres: Tensor | None = None,
                  dgamma: Tensor | None = None, dbeta: Tensor | None = None, accumulate: bool = False):
    """-> (dx f32 like x, dgamma, dbeta); ``dres`` (f32, x's shape) is added into dx."""
    _need_cuda(x, dy)
    D = x.shape[-1]
    if x.dtype != torch.float32 or x.stride(-1) != 1 or not dy.is_contiguous() or dy.shape != x.shape:
        raise ValueError("layernorm_bwd: x f32 with unit last stride, dy contiguous of the same shape")
    x2 = x.reshape(-1, D)
    rows = x2.shape[0]
    dx = torch.empty((rows, D), device=x.device, dtype=torch.float32)
    r2 = None
    if dres is not None:
        if dres.dtype != torch.float32 or dres.shape != x.shape or dres.stride(-1) != 1:
            raise ValueError("layernorm_bwd: dres must be f32 of x's shape")
        r2 = dres.reshape(-1, D)
    dgamma = torch.empty(D, device=x.device, dtype=torch.float32) if dgamma is None else dgamma
    dbeta = torch.empty(D, device=x.device, dtype=torch.float32) if dbeta is None else dbeta
    ws, nbytes = _colreduce_ws(rows, D, 2, x.device)
    check(_lib.load().gdl_layernorm_bwd(_p(x2), x2.stride(0), _p(dy), dt(dy), _p(_f32vec(gamma, D, "gamma")), _p(r2),
                                        0 if r2 is None else r2.stride(0), _p(dx), D, rows, D, eps, _p(dgamma),
                                        _p(dbeta), int(accumulate), _p(ws), nbytes, _stream()), "gdl_layernorm_bwd")
    return dx.reshape(x.shape), dgamma, dbeta


def colsum(x: Tensor, out: Tensor | None = None, accumulate: bool = False) -> Tensor:
    """out[c] (+)= sum over all leading dims of x[..., c] (bias gradients)."""
    _need_cuda(x)
    Cc = x.shape[-1]
    x2 = x.reshape(-1, Cc)
    if x2.stride(1) != 1:
        raise ValueError("colsum: unit channel stride expected")
    if out is None:
        out = torch.empty(Cc, device=x.device, dtype=torch.float32)
    ws, nbytes = _colreduce_ws(x2.shape[0], Cc, 1, x.device)
    check(_lib.load().gdl_colsum(_p(x2), dt(x2), x2.shape[0], Cc, x2.stride(0), _p(out), int(accumulate), _p(ws),
                                 nbytes, _stream()), "gdl_colsum")
    return out


def layerscale_bwd(g: Tensor, z: Tensor | None, gamma: Tensor | None, batch_scale: Tensor | None,
                   out_dtype: torch.dtype, dgamma: Tensor | None = None, accumulate: bool = False):
    """g f32 [B,N,C] -> (dz [B,N,C] in out_dtype = g*batch_scale*gamma, dgamma = sum g*batch_scale*z or None)."""
    _need_cuda(g)
    B, N, Cc = g.shape
    if g.dtype != torch.float32 or not g.is_contiguous():
        raise ValueError("layerscale_bwd: contiguous f32 gradient expected")
    dz = torch.empty((B, N, Cc), device=g.device, dtype=out_dtype)
    ws, nbytes = (None, 0)
    if gamma is not None:
        if z is None or not z.is_contiguous() or z.shape != g.shape:
            raise ValueError("layerscale_bwd: z must be contiguous and of g's shape")
        dgamma = torch.empty(Cc, device=g.device, dtype=torch.float32) if dgamma is None else dgamma
        ws, nbytes = _colreduce_ws(B * N, Cc, 1, g.device)
    zz = z if gamma is not None else None
    check(_lib.load().gdl_layerscale_bwd(_p(g), _p(zz), F32 if zz is None else dt(zz),
                                         _p(None if gamma is None else _f32vec(gamma, Cc, "gamma")),
                                         _p(None if batch_scale is None else _f32vec(batch_scale, B, "batch_scale")),
                                         B * N, N, Cc, _p(dz), dt(dz), _p(dgamma if gamma is not None else None),
                                         int(accumulate), _p(ws), nbytes, _stream()), "gdl_layerscale_bwd")
    return dz, (dgamma if gamma is not None else None)


def dwconv3x3_gelu_bwd(u: Tensor, dy: Tensor, w9: Tensor, bias: Tensor):
    """backward of gelu(dwconv3x3(u)+bias): -> (du like u, dw9 [9,C] f32, dbias [C] f32)."""
    _need_cuda(u, dy)
    if u.dim() != 4 or not u.is_contiguous() or not dy.is_contiguous() or dy.shape != u.shape or dy.dtype != u.dtype:
        raise ValueError("dwconv3x3_gelu_bwd: contiguous NHWC u and dy of one dtype expected")
    B, H, W, Cc = u.shape
    dpre = torch.empty_like(u)
    dw9 = torch.empty((9, Cc), device=u.device, dtype=torch.float32)
    db = torch.empty(Cc, device=u.device, dtype=torch.float32)
    ws, nbytes = _colreduce_ws(B * H * W, Cc, 10, u.device)
    check(_lib.load().gdl_dwconv3x3_gelu_bwd(_p(u), _p(dy), dt(u), B, H, W, Cc, _p(w9), _p(_f32vec(bias, Cc, "bias")),
                                             _p(dpre), _p(dw9), _p(db), 0, _p(ws), nbytes, _stream()),
          "gdl_dwconv3x3_gelu_bwd")
    zero_b = torch.zeros(Cc, device=u.device, dtype=torch.float32)
    du = dwconv3x3(dpre, w9.flip(0).contiguous(), zero_b, False)
    return du, dw9, db


# ------------------------------------------------------------------ dynamic (channel-adaptive) SegFormer stem
def _f32c(t: Tensor, shape: tuple, name: str) -> Tensor:
    if t.dtype != torch.float32 or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous f32 tensor of shape {tuple(shape)}, got {tuple(t.shape)} {t.dtype}")
    return t


def chan_weights(pos: Tensor, W0: Tensor, b0: Tensor, W2: Tensor, b2: Tensor, W1: Tensor, b1: Tensor):
    """weight_gen + the position half of channel_attention[0] (mix_transformer.py:781-801).
    pos [C, PD]; W1 = channel_attention.0.weight as [H1, E + PD].  -> (hid [C,HD], cw [C,E], hb [C,H1])"""
    _need_cuda(pos, W0, W2, W1)
    Cn, PD = pos.shape
    HD, E, H1 = W0.shape[0], W2.shape[0], W1.shape[0]
    _f32c(pos, (Cn, PD), "pos"), _f32c(W0, (HD, PD), "weight_gen.0.weight"), _f32c(W2, (E, HD), "weight_gen.2.weight")
    _f32c(W1, (H1, E + PD), "channel_attention.0.weight")
    hid = torch.empty((Cn, HD), device=pos.device, dtype=torch.float32)
    cw = torch.empty((Cn, E), device=pos.device, dtype=torch.float32)
    hb = torch.empty((Cn, H1), device=pos.device, dtype=torch.float32)
    w1b = W1[:, E:]
    check(_lib.load().gdl_chan_weights_fwd(_p(pos), Cn, PD, HD, E, H1, _p(W0), _p(_f32vec(b0, HD, "weight_gen.0.bias")),
                                           _p(W2), _p(_f32vec(b2, E, "weight_gen.2.bias")), _p(w1b), W1.stride(0),
                                           _p(_f32vec(b1, H1, "channel_attention.0.bias")), _p(hid), _p(cw), _p(hb),
                                           _stream()), "gdl_chan_weights_fwd")
    return hid, cw, hb


def chan_weights_bwd(pos: Tensor, W2: Tensor, hid: Tensor, cw: Tensor, dcw: Tensor, dhb: Tensor):
    """-> (dW0 [HD,PD], db0, dW2 [E,HD], db2, dW1b [H1,PD], db1)"""
    _need_cuda(pos, dcw, dhb)
    Cn, PD = pos.shape
    HD, E, H1 = hid.shape[1], cw.shape[1], dhb.shape[1]
    _f32c(dcw, (Cn, E), "dcw"), _f32c(dhb, (Cn, H1), "dhb"), _f32c(W2, (E, HD), "weight_gen.2.weight")
    dev = pos.device
    outs = [torch.empty(sh, device=dev, dtype=torch.float32) for sh in ((HD, PD), (HD,), (E, HD), (E,), (H1, PD), (H1,))]
    check(_lib.load().gdl_chan_weights_bwd(_p(pos), Cn, PD, HD, E, H1, _p(W2), _p(hid), _p(cw), _p(dcw), _p(dhb),
                                           *[_p(o) for o in outs], _stream()), "gdl_chan_weights_bwd")
    return tuple(outs)


def chan_pool(conv: Tensor, cw: Tensor, W1: Tensor, hb: Tensor, w2: Tensor, b2s: float):
    """conv [B, C, P, E] f32 -> (agg [B, P, E] f32, attn [B, P, C]) (mix_transformer.py:823-853)."""
    _need_cuda(conv, cw, W1, hb, w2)
    B, Cn, P, E = conv.shape
    H1 = hb.shape[1]
    _f32c(conv, (B, Cn, P, E), "conv"), _f32c(cw, (Cn, E), "cw"), _f32c(hb, (Cn, H1), "hb")
    if W1.dtype != torch.float32 or W1.shape[0] != H1 or W1.shape[1] < E or W1.stride(1) != 1:
        raise ValueError("chan_pool: W1 must be f32 [H1, >= E] with unit column stride")
    agg = torch.empty((B, P, E), device=conv.device, dtype=torch.float32)
    attn = torch.empty((B, P, Cn), device=conv.device, dtype=torch.float32)
    check(_lib.load().gdl_chan_pool_fwd(_p(conv), B, Cn, P, E, H1, _p(cw), _p(W1), W1.stride(0), _p(hb),
                                        _p(_f32vec(w2, H1, "channel_attention.2.weight")), float(b2s), _p(agg), _p(attn),
                                        _stream()), "gdl_chan_pool_fwd")
    return agg, attn


def chan_pool_bwd(conv: Tensor, cw: Tensor, W1: Tensor, hb: Tensor, w2: Tensor, b2s: float, dagg: Tensor):
    """-> (dconv like conv, dW1a [H1,E], dhb [C,H1], dw2 [H1], dcw [C,E])"""
    _need_cuda(conv, dagg)
    B, Cn, P, E = conv.shape
    H1 = hb.shape[1]
    _f32c(dagg, (B, P, E), "dagg")
    lib = _lib.load()
    nbytes = lib.gdl_chan_pool_workspace(B, Cn, P, E, H1)
    ws = torch.empty(max(nbytes, 4) // 4, device=conv.device, dtype=torch.float32)
    dconv = torch.empty_like(conv)
    grads = torch.empty(H1 * E + Cn * H1 + H1 + Cn * E, device=conv.device, dtype=torch.float32)
    check(lib.gdl_chan_pool_bwd(_p(conv), B, Cn, P, E, H1, _p(cw), _p(W1), W1.stride(0), _p(hb), _p(w2), float(b2s),
                                _p(dagg), _p(dconv), _p(grads), _p(ws), nbytes, _stream()), "gdl_chan_pool_bwd")
    o1, o2, o3 = H1 * E, H1 * E + Cn * H1, H1 * E + Cn * H1 + H1
    return dconv, grads[:o1].view(H1, E), grads[o1:o2].view(Cn, H1), grads[o2:o3], grads[o3:].view(Cn, E)


def col2im(cols: Tensor, B: int, Ho: int, Wo: int, R: int, S: int, Cc: int, stride: int, pad: int, H: int, W: int,
           out_dtype: torch.dtype) -> Tensor:
    """cols [B*Ho*Wo, R*S*C] -> dx NHWC [B,H,W,C] (data gradient of a strided conv)."""
    _need_cuda(cols)
    if not cols.is_contiguous() or cols.numel() != B * Ho * Wo * R * S * Cc:
        raise ValueError("col2im: contiguous [B*Ho*Wo, R*S*C] expected")
    dx = torch.empty((B, H, W, Cc), device=cols.device, dtype=out_dtype)
    check(_lib.load().gdl_col2im(_p(cols), dt(cols), B, Ho, Wo, R, S, Cc, stride, pad, H, W, _p(dx), dt(dx),
                                 dx.stride(0), dx.stride(1), dx.stride(2), _stream()), "gdl_col2im")
    return dx


def _bgemm(inp: Tensor, in_sW: int, in_sZ: tuple[int, int], w: Tensor, w_sN: int, w_sZ: tuple[int, int], out: Tensor,
           out_sW: int, out_sZ: tuple[int, int], M: int, K: int, N: int, nz: int, nz_inner: int, alpha: float) -> None:
    """out[z][m, n] = alpha * sum_k in[z][m, k] * w[z][n, k] over nz = (z0, z1) problems."""
    a = ConvArgs()
    a.inp, a.dtype = inp.data_ptr(), dt(inp)
    a.B, a.H, a.W, a.C = 1, 1, M, K
    a.in_sB, a.in_sH, a.in_sW = M * in_sW, M * in_sW, in_sW
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad = 1, M, 1, 1, 1, 0
    a.w, a.w_sN, a.N = w.data_ptr(), w_sN, N
    a.out, a.out_dtype = out.data_ptr(), dt(out)
    a.out_sB, a.out_sH, a.out_sW = M * out_sW, M * out_sW, out_sW
    a.alpha, a.act = alpha, ACT_NONE
    a.nz, a.nz_inner = nz, nz_inner
    a.in_sZ0, a.in_sZ1 = in_sZ
    a.w_sZ0, a.w_sZ1 = w_sZ
    a.out_sZ0, a.out_sZ1 = out_sZ
    batched_gemm_raw(a)


def _bwgrad(x: Tensor, x_sW: int, x_sZ: tuple[int, int], dy: Tensor, dy_sW: int, dy_sZ: tuple[int, int], dw: Tensor,
            P: int, Cc: int, N: int, nz: int, nz_inner: int) -> None:
    """dw[z][n, c] = sum_p dy[z][p, n] * x[z][p, c] (f32), dw dense [nz, N, C]."""
    a = WgradArgs()
    a.inp, a.dy, a.dtype = x.data_ptr(), dy.data_ptr(), dt(x)
    a.B, a.H, a.W, a.C = 1, 1, P, Cc
    a.in_sB, a.in_sH, a.in_sW = P * x_sW, P * x_sW, x_sW
    a.Ho, a.Wo, a.R, a.S, a.stride, a.pad, a.N = 1, P, 1, 1, 1, 0, N
    a.dy_sB, a.dy_sH, a.dy_sW = P * dy_sW, P * dy_sW, dy_sW
    a.dw, a.dw_sN, a.accumulate = dw.data_ptr(), Cc, 0
    a.nz, a.nz_inner = nz, nz_inner
    a.in_sZ0, a.in_sZ1 = x_sZ
    a.dy_sZ0, a.dy_sZ1 = dy_sZ
    a.dw_sZ0, a.dw_sZ1 = nz_inner * N * Cc, N * Cc
    lib = _lib.load()
    nbytes = lib.gdl_conv_wgrad_workspace(C.byref(a))      # non-zero only for nz == 1 (split-K)
    ws = torch.empty(max(nbytes, 4) // 4, device=x.device, dtype=torch.float32)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(lib.gdl_conv_wgrad(C.byref(a), _stream()), "gdl_conv_wgrad(batched)")


def attention_bwd(q: Tensor, k: Tensor, v: Tensor, do: Tensor, num_heads: int, dq: Tensor, dk: Tensor,
                  dv: Tensor, o: Tensor | None = None, lse: Tensor | None = None) -> None:
    """Backward of softmax(q k^T / sqrt(hd)) v.  q/do/dq [B,Nq,D], k/v/dk/dv [B,Nkv,D]; all may be strided
    slices of packed qkv / kv tensors (unit channel stride); dq/dk/dv are written in place.

    With the forward's output ``o`` and ``lse`` (bf16, head_dim 64) the fused kernels run (nothing of size Nq x Nkv in
    HBM); otherwise -- the exact-f32 parity path and odd head dims -- the materialised path below.

    The probabilities are recomputed (the forward is the fused flash kernel and keeps none):
    S = scale q k^T -> P = softmax(S); dP = dO V^T; dS = P*(dP - rowsum(dP*P))*scale;
    dQ = dS K (batched GEMM against K^T); dK = dS^T Q, dV = P^T dO (batched weight-gradient kernels)."""
    B, Nq, Nkv, D, hd = _attn_check(q, k, v, num_heads)
    if o is not None and lse is not None and flash_ok(q, num_heads):
        attention_flash_bwd(q, k, v, o, do, lse, num_heads, dq, dk, dv)
        return
    cdt = q.dtype
    al = 4 if cdt == torch.float32 else 8
    if hd % al != 0:
        raise ValueError(f"attention_bwd: head_dim {hd} must be a multiple of {al} for {cdt}")
    for t, n in ((do, Nq), (dq, Nq), (dk, Nkv), (dv, Nkv)):
        if t.shape != (B, n, D) or t.dtype != cdt or t.stride(2) != 1:
            raise ValueError("attention_bwd: gradient tensor shape / dtype / stride mismatch")
    Hh = num_heads
    npad = (Nkv + 63) // 64 * 64
    lib = _lib.load()
    nz = B * Hh
    sc_z = (Hh * Nq * npad, Nq * npad)
    scale = float(hd) ** -0.5
    P = torch.empty((B, Hh, Nq, npad), device=q.device, dtype=cdt)
    _bgemm(q, q.stride(1), (q.stride(0), hd), k, k.stride(1), (k.stride(0), hd), P, npad, sc_z, Nq, hd, Nkv, nz, Hh,
           scale)
    check(lib.gdl_softmax_rows(_p(P), _p(P), dt(P), nz * Nq, Nkv, npad, _stream()), "gdl_softmax_rows")
    dS = torch.empty_like(P)
    _bgemm(do, do.stride(1), (do.stride(0), hd), v, v.stride(1), (v.stride(0), hd), dS, npad, sc_z, Nq, hd, Nkv, nz, Hh,
           1.0)
    check(lib.gdl_softmax_bwd_rows(_p(P), _p(dS), _p(dS), dt(P), nz * Nq, Nkv, npad, scale, _stream()),
          "gdl_softmax_bwd_rows")
    kt = _v_transposed(k, Hh, hd, npad)                                    # K^T [B,H,hd,npad]
    _bgemm(dS, npad, sc_z, kt, npad, (Hh * hd * npad, hd * npad), dq, dq.stride(1), (dq.stride(0), hd), Nq, npad, hd,
           nz, Hh, 1.0)
    dkv32 = torch.empty((2, B, Hh, npad, hd), device=q.device, dtype=torch.float32)
    _bwgrad(q, q.stride(1), (q.stride(0), hd), dS, npad, sc_z, dkv32[0], Nq, hd, npad, nz, Hh)
    _bwgrad(do, do.stride(1), (do.stride(0), hd), P, npad, sc_z, dkv32[1], Nq, hd, npad, nz, Hh)
    for src, dst in ((dkv32[0], dk), (dkv32[1], dv)):
        # [B,H,n,hd] f32 -> [B,n,H*hd] slice in the compute dtype: a strided copy-cast (identity resample)
        bilinear(src.permute(0, 2, 1, 3)[:, :Nkv], (Nkv, Hh),
                 out=dst.as_strided((B, Nkv, Hh, hd), (dst.stride(0), dst.stride(1), hd, 1)))


# ------------------------------------------------------------------ ResNet / UNet++ pieces
def maxpool3x3s2(x: Tensor) -> Tensor:
    """F.max_pool2d(kernel 3, stride 2, padding 1) on NHWC."""
    _need_cuda(x)
    x4 = _nhwc4(x, "maxpool x")
    B, H, W, Cc = x4.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), device=x.device, dtype=x.dtype)
    check(_lib.load().gdl_maxpool3x3s2_fwd(_p(x4), dt(x4), B, H, W, Cc, x4.stride(0), x4.stride(1), x4.stride(2), _p(out),
                                           out.stride(0), out.stride(1), out.stride(2), _stream()), "gdl_maxpool3x3s2_fwd")
    return out


def maxpool3x3s2_bwd(x: Tensor, dout: Tensor) -> Tensor:
    x4, d4 = _nhwc4(x, "maxpool_bwd x"), _nhwc4(dout, "maxpool_bwd dout")
    B, H, W, Cc = x4.shape
    din = torch.empty((B, H, W, Cc), device=x.device, dtype=x.dtype)
    check(_lib.load().gdl_maxpool3x3s2_bwd(_p(x4), _p(d4), _p(din), dt(x4), B, H, W, Cc, x4.stride(0), x4.stride(1),
                                           x4.stride(2), d4.stride(0), d4.stride(1), d4.stride(2), din.stride(0),
                                           din.stride(1), din.stride(2), _stream()), "gdl_maxpool3x3s2_bwd")
    return din


def nearest2x(x: Tensor, out: Tensor | None = None) -> Tensor:
    """F.interpolate(scale_factor=2, mode='nearest') on NHWC; ``out`` may be a channel slice of a concat buffer."""
    _need_cuda(x)
    x4 = _nhwc4(x, "nearest2x x")
    B, H, W, Cc = x4.shape
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cc), device=x.device, dtype=x.dtype)
    o4 = _nhwc4(out, "nearest2x out")
    if tuple(o4.shape) != (B, 2 * H, 2 * W, Cc) or o4.dtype != x4.dtype:
        raise ValueError("nearest2x: out shape / dtype mismatch")
    check(_lib.load().gdl_nearest2x_fwd(_p(x4), dt(x4), B, H, W, Cc, x4.stride(0), x4.stride(1), x4.stride(2), _p(o4),
                                        o4.stride(0), o4.stride(1), o4.stride(2), _stream()), "gdl_nearest2x_fwd")
    return out


def nearest2x_bwd(dout: Tensor) -> Tensor:
    d4 = _nhwc4(dout, "nearest2x_bwd dout")
    B, H2, W2, Cc = d4.shape
    din = torch.empty((B, H2 // 2, W2 // 2, Cc), device=dout.device, dtype=dout.dtype)
    check(_lib.load().gdl_nearest2x_bwd(_p(d4), dt(d4), B, H2 // 2, W2 // 2, Cc, d4.stride(0), d4.stride(1), d4.stride(2),
                                        _p(din), din.stride(0), din.stride(1), din.stride(2), _stream()),
          "gdl_nearest2x_bwd")
    return din


def add_relu(a: Tensor, b: Tensor) -> Tensor:
    if a.shape != b.shape or a.dtype != b.dtype or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("add_relu: two contiguous tensors of one shape / dtype expected")
    out = torch.empty_like(a)
    check(_lib.load().gdl_add_relu(_p(a), _p(b), _p(out), dt(a), a.numel(), _stream()), "gdl_add_relu")
    return out


def pad_channels(x: Tensor, cpad: int, out_dtype: torch.dtype) -> Tensor:
    """Dense [..., C] -> [..., cpad] in out_dtype with zero-filled extra channels."""
    _need_cuda(x)
    if not x.is_contiguous():
        raise ValueError("pad_channels: contiguous input expected")
    Cc = x.shape[-1]
    out = torch.empty((*x.shape[:-1], cpad), device=x.device, dtype=out_dtype)
    check(_lib.load().gdl_pad_channels(_p(x), dt(x), x.numel() // Cc, Cc, _p(out), dt(out), cpad, _stream()),
          "gdl_pad_channels")
    return out


def relu_bwd(y: Tensor, dy: Tensor) -> Tensor:
    if y.shape != dy.shape or y.dtype != dy.dtype or not y.is_contiguous() or not dy.is_contiguous():
        raise ValueError("relu_bwd: two contiguous tensors of one shape / dtype expected")
    dx = torch.empty_like(y)
    check(_lib.load().gdl_relu_bwd(_p(y), _p(dy), _p(dx), dt(y), y.numel(), _stream()), "gdl_relu_bwd")
    return dx


# ------------------------------------------------------------------ DOFA patch embed helpers
def patchify(img: Tensor, P: int, pad: int, gh: int, gw: int, kpad: int,
             out_dtype: torch.dtype, stride: int | None = None) -> Tensor:
    _need_cuda(img)
    if img.dtype != torch.float32 or not img.is_contiguous():
        raise ValueError("patchify: image must be contiguous f32 NCHW")
    B, Cc, H, W = img.shape
    cols = torch.empty((B * gh * gw, kpad), device=img.device, dtype=out_dtype)
    check(_lib.load().gdl_patchify(_p(img), B, Cc, H, W, P, P if stride is None else stride, pad, gh, gw,
                                   _p(cols), dt(cols), kpad, _stream()), "gdl_patchify")
    return cols


def dofa_pack_kernel(g: Tensor, Cc: int, PP: int, D: int, scaler: float, kpad: int,
                     out_dtype: torch.dtype) -> Tensor:
    if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != Cc * PP * D:
        raise ValueError("dofa_pack_kernel: g must be contiguous f32 [C, P*P*D]")
    out = torch.empty((D, kpad), device=g.device, dtype=out_dtype)
    check(_lib.load().gdl_dofa_pack_kernel(_p(g), Cc, PP, D, scaler, _p(out), dt(out), kpad,
                                           _stream()), "gdl_dofa_pack_kernel")
    return out


_OMEGA: dict = {}


def dofa_unpack_grad(dw: Tensor, Cc: int, PP: int, D: int, scaler: float) -> Tensor:
    """f32 [D, Kpad] gradient of the packed patch-embed weight -> [C, PP*D] gradient of the generated kernel."""
    if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.shape[0] != D:
        raise ValueError("dofa_unpack_grad: contiguous f32 [D, Kpad] expected")
    dg = torch.empty((Cc, PP * D), device=dw.device, dtype=torch.float32)
    check(_lib.load().gdl_dofa_unpack_grad(_p(dw), Cc, PP, D, scaler, dw.shape[1], _p(dg), _stream()),
          "gdl_dofa_unpack_grad")
    return dg


def sincos_embed(pos: Tensor, D: int) -> Tensor:
    """position_embedding(D, pos) (dofa_v2.py:9-35); the frequency table is a host constant."""
    _need_cuda(pos)
    pos = pos.reshape(-1).contiguous().float()
    key = (D, pos.device)
    if key not in _OMEGA:
        omega = torch.arange(D // 2, dtype=torch.float32)
        omega /= D / 2.0
        _OMEGA[key] = (1.0 / 10000**omega).to(pos.device)
    out = torch.empty((pos.numel(), D), device=pos.device, dtype=torch.float32)
    check(_lib.load().gdl_sincos_embed(_p(pos), _p(_OMEGA[key]), pos.numel(), D, _p(out), _stream()),
          "gdl_sincos_embed")
    return out


def bn_fold(gamma: Tensor, beta: Tensor, mean: Tensor, var: Tensor, eps: float):
    Cc = gamma.numel()
    scale = torch.empty(Cc, device=gamma.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    check(_lib.load().gdl_bn_fold(_p(gamma), _p(beta), _p(mean), _p(var), eps, Cc, _p(scale),
                                  _p(shift), _stream()), "gdl_bn_fold")
    return scale, shift


def pack_dgrad(w: Tensor, N: int, T: int, Cc: int, out_dtype: torch.dtype) -> Tensor:
    """[N, T*C] forward weights -> [C, T*N] flipped/transposed weights for the data gradient."""
    if not w.is_contiguous() or w.numel() != N * T * Cc:
        raise ValueError("pack_dgrad: contiguous [N, T*C] weights expected")
    out = torch.empty((Cc, T * N), device=w.device, dtype=out_dtype)
    check(_lib.load().gdl_pack_dgrad(_p(w), dt(w), N, T, Cc, _p(out), dt(out), _stream()),
          "gdl_pack_dgrad")
    return out


# ------------------------------------------------------------------ elementwise
def cast(x: Tensor, dtype: torch.dtype, out: Tensor | None = None) -> Tensor:
    _need_cuda(x)
    if not x.is_contiguous():
        raise ValueError("cast: contiguous input required")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(_lib.load().gdl_cast(_p(x), dt(x), _p(out), dt(out), x.numel(), _stream()), "gdl_cast")
    return out


def scale_f32(x: Tensor, s: float) -> Tensor:
    out = torch.empty_like(x)
    check(_lib.load().gdl_scale_f32(_p(x), _p(out), x.numel(), s, _stream()), "gdl_scale_f32")
    return out


def add_rows(a: Tensor, b: Tensor | None, out: Tensor, rows: int) -> Tensor:
    """out[r,:] = a[r % a_rows,:] + (b[r % b_rows,:] if b is given); 2-D f32 tensors."""
    D = a.shape[-1]
    check(_lib.load().gdl_add_rows(_p(a), a.shape[0], a.stride(0), _p(b),
                                   0 if b is None else b.shape[0], 0 if b is None else b.stride(0),
                                   _p(out), out.stride(0), rows, D, _stream()), "gdl_add_rows")
    return out


def normalize_u8(u8: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """uint8 NCHW -> (x/255 - mean[c]) / std[c] f32 (utils/tensors.py:10-35)."""
    _need_cuda(u8)
    if u8.dtype != torch.uint8 or not u8.is_contiguous():
        raise ValueError("normalize_u8: contiguous uint8 NCHW expected")
    B, Cc, H, W = u8.shape
    out = torch.empty((B, Cc, H, W), device=u8.device, dtype=torch.float32)
    check(_lib.load().gdl_normalize_u8(_p(u8), _p(out), B, Cc, H * W, _p(_f32vec(mean, Cc, "mean")),
                                       _p(_f32vec(std, Cc, "std")), _stream()), "gdl_normalize_u8")
    return out


_RAW_KIND = {torch.uint8: 0, torch.uint16: 1, torch.int16: 2, torch.float32: 3}


def normalize_raw(raw: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Raw NCHW samples (uint8 / uint16 / int16 / f32) -> (float(x)/255 - mean[c]) / std[c] f32
    (datasets/wds_dataset.py:230-236 + utils/tensors.py:10-35)."""
    _need_cuda(raw)
    if raw.dtype not in _RAW_KIND or not raw.is_contiguous() or raw.dim() != 4:
        raise ValueError(f"normalize_raw: contiguous NCHW uint8/uint16/int16/float32 expected, got {raw.dtype}")
    B, Cc, H, W = raw.shape
    out = torch.empty((B, Cc, H, W), device=raw.device, dtype=torch.float32)
    check(_lib.load().gdl_normalize_raw(_p(raw), _RAW_KIND[raw.dtype], _p(out), B, Cc, H * W,
                                        _p(_f32vec(mean, Cc, "mean")), _p(_f32vec(std, Cc, "std")), _stream()),
          "gdl_normalize_raw")
    return out


def augment(img: Tensor, mask: Tensor | None, params: Tensor, mean: Tensor | None = None,
            std: Tensor | None = None) -> tuple[Tensor, Tensor | None]:
    """Per-sample flip / rot90 / resized-crop of an NCHW tile (+ its int64 mask), optionally fused with the
    /255 + standardise of a raw tile.  ``params``: f32 [B, 8] = {kind, k, y0, x0, h, w, -, -} on the device."""
    _need_cuda(img, params)
    if img.dtype not in _RAW_KIND or not img.is_contiguous() or img.dim() != 4:
        raise ValueError("augment: contiguous NCHW uint8/uint16/int16/float32 tile expected")
    B, Cc, H, W = img.shape
    if params.shape != (B, 8) or params.dtype != torch.float32 or not params.is_contiguous():
        raise ValueError("augment: params must be a contiguous f32 [B, 8] tensor")
    out = torch.empty((B, Cc, H, W), device=img.device, dtype=torch.float32)
    m2 = om = None
    if mask is not None:
        if mask.dtype != torch.int64 or mask.numel() != B * H * W:
            raise ValueError("augment: int64 mask of B*H*W elements expected")
        m2 = mask.contiguous()
        om = torch.empty_like(m2)
    check(_lib.load().gdl_augment(_p(img), _RAW_KIND[img.dtype], _p(out), _p(m2), _p(om), B, Cc, H, W,
                                  _p(None if mean is None else _f32vec(mean, Cc, "mean")),
                                  _p(None if std is None else _f32vec(std, Cc, "std")), _p(params), _stream()),
          "gdl_augment")
    return out, om


def scale_outer(x: Tensor, s: Tensor) -> Tensor:
    """x[o, ...] *= s[o] in place."""
    if not x.is_contiguous():
        raise ValueError("scale_outer: contiguous tensor required")
    outer = x.shape[0]
    check(_lib.load().gdl_scale_outer(_p(x), dt(x), _p(_f32vec(s, outer, "s")), outer,
                                      x.numel() // outer, _stream()), "gdl_scale_outer")
    return x


# ------------------------------------------------------------------ classifier tail
def head_1x1(feat: Tensor, w: Tensor, bias: Tensor | None, chan_scale: Tensor | None = None) -> Tensor:
    """NHWC features -> f32 NHWC logits [B,H,W,K] (K <= 16)."""
    f4 = _nhwc4(feat, "head feat")
    B, H, W, Cc = f4.shape
    P, _, sP = _pix(f4, "head feat")
    K = w.shape[0]
    w2 = w.reshape(K, Cc)
    if w2.dtype != torch.float32 or not w2.is_contiguous():
        raise ValueError("head_1x1: weight must be contiguous f32 [K, C]")
    out = torch.empty((B, H, W, K), device=feat.device, dtype=torch.float32)
    check(_lib.load().gdl_head_1x1(_p(f4), dt(f4), P, Cc, sP, _p(w2), _p(bias), _p(chan_scale),
                                   H * W, _p(out), K, _stream()), "gdl_head_1x1")
    return out


def head_1x1_bwd(feat: Tensor, dlog: Tensor, w: Tensor, chan_scale: Tensor | None,
                 need_dfeat: bool = True):
    f4 = _nhwc4(feat, "head feat")
    B, H, W, Cc = f4.shape
    P, _, sP = _pix(f4, "head feat")
    K = w.shape[0]
    w2 = w.reshape(K, Cc)
    dfeat = torch.empty((B, H, W, Cc), device=feat.device, dtype=feat.dtype) if need_dfeat else None
    dw = torch.empty((K, Cc), device=feat.device, dtype=torch.float32)
    db = torch.empty(K, device=feat.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.gdl_head_1x1_bwd_workspace(P, Cc, K)
    ws = torch.empty(nbytes // 4, device=feat.device, dtype=torch.float32)
    check(lib.gdl_head_1x1_bwd(_p(f4), dt(f4), _p(dlog), P, Cc, sP, _p(w2), _p(chan_scale), H * W,
                               _p(dfeat), Cc, _p(dw), _p(db), K, _p(ws), nbytes, _stream()),
          "gdl_head_1x1_bwd")
    return dfeat, dw, db


def upsample_logits(x: Tensor, size: tuple[int, int]) -> Tensor:
    """f32 NHWC [B,Hi,Wi,K] -> NCHW f32 [B,K,Ho,Wo] (bilinear, align_corners=False)."""
    B, Hi, Wi, K = x.shape
    out = torch.empty((B, K, size[0], size[1]), device=x.device, dtype=torch.float32)
    check(_lib.load().gdl_upsample_logits(_p(x), B, Hi, Wi, K, _p(out), size[0], size[1], _stream()),
          "gdl_upsample_logits")
    return out


def upsample_logits_bwd(dout: Tensor, in_size: tuple[int, int]) -> Tensor:
    if dout.dtype != torch.float32 or not dout.is_contiguous():
        raise ValueError("upsample_logits_bwd: contiguous f32 NCHW grad expected")
    B, K, Ho, Wo = dout.shape
    din = torch.empty((B, in_size[0], in_size[1], K), device=dout.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.gdl_upsample_logits_bwd_workspace(B, K, in_size[0], Wo)
    ws = torch.empty(nbytes // 4, device=dout.device, dtype=torch.float32)
    check(lib.gdl_upsample_logits_bwd(_p(dout), B, Ho, Wo, K, _p(din), in_size[0], in_size[1], _p(ws),
                                      nbytes, _stream()), "gdl_upsample_logits_bwd")
    return din


def softmax_argmax(logits: Tensor) -> Tensor:
    """softmax(dim=1).argmax(dim=1) on NCHW f32 logits -> int64 [B,H,W]."""
    _need_cuda(logits)
    if logits.dtype != torch.float32 or not logits.is_contiguous():
        raise ValueError("softmax_argmax: contiguous f32 NCHW logits expected")
    B, K, H, W = logits.shape
    mask = torch.empty((B, H, W), device=logits.device, dtype=torch.int64)
    check(_lib.load().gdl_softmax_argmax(_p(logits), B, K, H * W, _p(mask), _stream()),
          "gdl_softmax_argmax")
    return mask


def upsample_argmax(low: Tensor, size: tuple[int, int]) -> Tensor:
    """softmax(dim=1).argmax(dim=1) of bilinear(low -> size) for the head's NHWC f32 map [B, h, w, K] -> int64 [B, H, W], without the
    resized logits (gdl_upsample_argmax; the same mask as upsample_logits + softmax_argmax)."""
    _need_cuda(low)
    if low.dtype != torch.float32 or not low.is_contiguous() or low.dim() != 4 or not 2 <= low.shape[3] <= 16:
        raise ValueError("upsample_argmax: contiguous f32 NHWC logits [B, h, w, K] with 2..16 classes expected")
    B, Hi, Wi, K = low.shape
    mask = torch.empty((B, int(size[0]), int(size[1])), device=low.device, dtype=torch.int64)
    check(_lib.load().gdl_upsample_argmax(_p(low), B, Hi, Wi, K, _p(mask), int(size[0]), int(size[1]), _stream()), "gdl_upsample_argmax")
    return mask


def class_probs(logits: Tensor) -> Tensor:
    """NCHW f32 logits -> softmax(dim=1) (or sigmoid for one class) probabilities."""
    _need_cuda(logits)
    if logits.dtype != torch.float32 or not logits.is_contiguous() or logits.dim() != 4:
        raise ValueError("class_probs: contiguous f32 NCHW logits expected")
    B, K, H, W = logits.shape
    out = torch.empty_like(logits)
    check(_lib.load().gdl_class_probs(_p(logits), B, K, H * W, _p(out), _stream()), "gdl_class_probs")
    return out


def iou_counts(pred: Tensor, target: Tensor, num_classes: int) -> Tensor:
    """int64 [B, 3, K]: per sample and class (intersection, |pred==k|, |target==k|); exact integer counts."""
    _need_cuda(pred, target)
    if pred.dtype != torch.int64 or target.dtype != torch.int64 or pred.shape != target.shape:
        raise ValueError("iou_counts: int64 index tensors of one shape expected")
    B = pred.shape[0]
    p2, t2 = pred.reshape(B, -1).contiguous(), target.reshape(B, -1).contiguous()
    counts = torch.empty((B, 3, num_classes), device=pred.device, dtype=torch.int64)
    check(_lib.load().gdl_iou_counts(_p(p2), _p(t2), B, p2.shape[1], num_classes, _p(counts), _stream()),
          "gdl_iou_counts")
    return counts


def dice_loss_fwd(logits: Tensor, target: Tensor, eps: float = 1e-7):
    _need_cuda(logits, target)
    if logits.dtype != torch.float32 or not logits.is_contiguous():
        raise ValueError("dice_loss: contiguous f32 NCHW logits expected")
    if target.dtype != torch.int64 or not target.is_contiguous():
        raise ValueError("dice_loss: contiguous int64 target expected")
    B, K, H, W = logits.shape
    sums = torch.empty(3 * K, device=logits.device, dtype=torch.float32)
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.gdl_dice_loss_workspace(B, K, H * W)
    ws = torch.empty(nbytes // 4, device=logits.device, dtype=torch.float32)
    check(lib.gdl_dice_loss_fwd(_p(logits), _p(target), B, K, H * W, eps, _p(sums), _p(loss), _p(ws),
                                nbytes, _stream()), "gdl_dice_loss_fwd")
    return loss, sums


def dice_loss_bwd(logits: Tensor, target: Tensor, sums: Tensor, upstream: Tensor | None,
                  grad_scale: float = 1.0, eps: float = 1e-7, out: Tensor | None = None,
                  accumulate: bool = False) -> Tensor:
    B, K, H, W = logits.shape
    if out is None:
        out = torch.empty_like(logits)
    check(_lib.load().gdl_dice_loss_bwd(_p(logits), _p(target), B, K, H * W, eps, _p(sums),
                                        _p(upstream), grad_scale, _p(out), int(accumulate),
                                        _stream()), "gdl_dice_loss_bwd")
    return out


def dice_lowres_ok(low: Tensor, size: tuple[int, int]) -> bool:
    """Shapes gdl_dice_loss_lowres_* take: an upsample by at most 64 per direction (DOFA's auxiliary head: 16 x 16 -> 512 x 512), at
    most 16 classes."""
    if low.dim() != 4 or low.shape[3] > 16:
        return False
    hi, wi = low.shape[1], low.shape[2]
    return size[0] >= hi and size[1] >= wi and -(-size[0] // hi) <= 64 and -(-size[1] // wi) <= 64


def dice_loss_lowres_fwd(low: Tensor, target: Tensor, size: tuple[int, int], eps: float = 1e-7):
    """Dice(multiclass) of bilinear(low -> size) vs target [B, H, W] without the full-resolution logits: (loss, sums)."""
    _need_cuda(low, target)
    if low.dtype != torch.float32 or not low.is_contiguous() or low.dim() != 4:
        raise ValueError("dice_loss_lowres: contiguous f32 NHWC low-resolution logits [B, h, w, K] expected")
    B, Hi, Wi, K = low.shape
    if target.dtype != torch.int64 or not target.is_contiguous() or tuple(target.shape) != (B, size[0], size[1]):
        raise ValueError(f"dice_loss_lowres: contiguous int64 target [B, {size[0]}, {size[1]}] expected, got {tuple(target.shape)}")
    sums = torch.empty(3 * K, device=low.device, dtype=torch.float32)
    loss = torch.empty((), device=low.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.gdl_dice_loss_lowres_workspace(B, K, size[0], size[1])
    ws = torch.empty(nbytes // 4, device=low.device, dtype=torch.float32)
    check(lib.gdl_dice_loss_lowres_fwd(_p(low), _p(target), B, K, Hi, Wi, size[0], size[1], eps, _p(sums), _p(loss), _p(ws), nbytes,
                                       _stream()), "gdl_dice_loss_lowres_fwd")
    return loss, sums


def dice_loss_lowres_bwd(low: Tensor, target: Tensor, size: tuple[int, int], sums: Tensor, upstream: Tensor | None,
                         grad_scale: float = 1.0, eps: float = 1e-7) -> Tensor:
    B, Hi, Wi, K = low.shape
    dlow = torch.empty_like(low)
    lib = _lib.load()
    nbytes = lib.gdl_dice_loss_lowres_bwd_workspace(B, K, Hi, Wi, size[0], size[1])
    ws = torch.empty(nbytes // 4, device=low.device, dtype=torch.float32) if nbytes else None
    check(lib.gdl_dice_loss_lowres_bwd(_p(low), _p(target), B, K, Hi, Wi, size[0], size[1], eps, _p(sums), _p(upstream),
                                       grad_scale, _p(dlow), _p(ws), nbytes, _stream()), "gdl_dice_loss_lowres_bwd")
    return dlow


def dice_binary_loss_fwd(logits: Tensor, target: Tensor, eps: float = 1e-7):
    """smp DiceLoss(mode="binary"): logits [B,1,H,W] (or any shape) f32, target of the same numel, int64 0/1."""
    _need_cuda(logits, target)
    if logits.dtype != torch.float32 or not logits.is_contiguous():
        raise ValueError("dice_binary_loss: contiguous f32 logits expected")
    if target.dtype != torch.int64 or not target.is_contiguous() or target.numel() != logits.numel():
        raise ValueError("dice_binary_loss: contiguous int64 target with one entry per logit expected")
    total = logits.numel()
    sums = torch.empty(3, device=logits.device, dtype=torch.float32)
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.gdl_dice_loss_workspace(1, 1, total)
    ws = torch.empty(nbytes // 4, device=logits.device, dtype=torch.float32)
    check(lib.gdl_dice_binary_loss_fwd(_p(logits), _p(target), total, eps, _p(sums), _p(loss), _p(ws), nbytes,
                                       _stream()), "gdl_dice_binary_loss_fwd")
    return loss, sums


def dice_binary_loss_bwd(logits: Tensor, target: Tensor, sums: Tensor, upstream: Tensor | None,
                         grad_scale: float = 1.0, eps: float = 1e-7) -> Tensor:
    out = torch.empty_like(logits)
    check(_lib.load().gdl_dice_binary_loss_bwd(_p(logits), _p(target), logits.numel(), eps, _p(sums), _p(upstream),
                                               grad_scale, _p(out), 0, _stream()), "gdl_dice_binary_loss_bwd")
    return out


# ------------------------------------------------------------------ optimizer
def sumsq_accum(x: Tensor, acc: Tensor) -> None:
    check(_lib.load().gdl_sumsq(_p(x), x.numel(), _p(acc), _stream()), "gdl_sumsq")


def clip_coef(sumsq: Tensor, max_norm: float, coef: Tensor) -> None:
    check(_lib.load().gdl_clip_coef(_p(sumsq), max_norm, _p(coef), _stream()), "gdl_clip_coef")


def multi_sumsq(table: Tensor, acc: Tensor) -> None:
    """acc += sum of squares of every gradient chunk listed in the device table [nchunks, 5]."""
    check(_lib.load().gdl_multi_sumsq(_p(table), table.shape[0], _p(acc), _stream()), "gdl_multi_sumsq")


def multi_adam(table: Tensor, lr: float, b1: float, b2: float, eps: float, wd: float, step: int,
               clip: Tensor | None) -> None:
    bc1, bc2 = 1.0 - b1**step, 1.0 - b2**step
    check(_lib.load().gdl_multi_adam(_p(table), table.shape[0], lr, b1, b2, eps, wd, bc1, bc2, _p(clip),
                                     _stream()), "gdl_multi_adam")


def adam_tick(state: Tensor, b1: float, b2: float) -> None:
    """state[0] += 1 and the two bias corrections refreshed, on the device (capturable Adam, see gdl_adam_tick)."""
    check(_lib.load().gdl_adam_tick(_p(state), float(b1), float(b2), _stream()), "gdl_adam_tick")


def multi_adam_dev(table: Tensor, state: Tensor, clip: Tensor | None) -> None:
    check(_lib.load().gdl_multi_adam_dev(_p(table), table.shape[0], _p(state), _p(clip), _stream()), "gdl_multi_adam_dev")


def multi_repack(table: Tensor, total_tiles: int) -> None:
    """Rebuild the bf16 operands derived from 3x3 conv parameters (channel-slice, tap-major, data-gradient layouts) listed in
    ``table`` (device int64 [rows, 10], see gdl_multi_repack) in one launch."""
    check(_lib.load().gdl_multi_repack(_p(table), table.shape[0], int(total_tiles), _stream()), "gdl_multi_repack")


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, b1: float, b2: float,
              eps: float, wd: float, step: int, clip: Tensor | None) -> None:
    bc1, bc2 = 1.0 - b1**step, 1.0 - b2**step
    check(_lib.load().gdl_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, wd, bc1,
                                    bc2, _p(clip), _stream()), "gdl_adam_step")
