"""CPU-only tests: the C-ABI library builds/loads and exports every symbol include/gdlhip.h
declares (no compute calls without a GPU), and the host-side mirror of the reference's
interface (class names, constructor arguments, state-dict keys, error behaviour)."""

import re
import sys
from pathlib import Path

import pytest
import torch

import oracle

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from gdlhip import _lib
    if not _lib.LIB_PATH.exists():
        import __graft_entry__ as ge
        ge.build()
    return _lib.load()


def _declared():
    text = (ROOT / "include" / "gdlhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gdl_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from gdlhip import _lib
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gdlhip.h but not exported by libgdlhip.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names, "binding and header disagree"
    assert lib.gdl_version() >= 100
    assert lib.gdl_last_error() is not None


def test_struct_layout_matches_header():
    """ctypes mirrors must have the header's field order and count."""
    from gdlhip import _lib
    text = (ROOT / "include" / "gdlhip.h").read_text()
    for cname, struct in (("gdl_conv_args", _lib.ConvArgs), ("gdl_wgrad_args", _lib.WgradArgs)):
        body = re.search(r"typedef struct \{([^}]*)\} " + cname + ";", text).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
            fields += [x.strip().lstrip("*").strip() for x in names.split(",")]
        mine = [f[0] for f in struct._fields_]
        assert len(mine) == len(fields), (cname, mine, fields)
        assert [m.replace("inp", "in") for m in mine] == fields


def test_ops_refuse_cpu_tensors(lib):
    from gdlhip import ops
    with pytest.raises(ValueError):
        ops.conv_gemm(torch.zeros(1, 2, 2, 64), torch.zeros(64, 64))
    with pytest.raises(ValueError):
        ops.layernorm(torch.zeros(2, 64), torch.ones(64), torch.zeros(64), 1e-5, torch.float32)


def test_state_dict_and_ctor_contract():
    """Same keys/shapes as the reference (332 tensors for DOFA-base)."""
    from geo_deep_learning.models.segmentation.dofa import DOFASegmentationModel
    from geo_deep_learning.models.heads.segmentation_head import SegmentationOutput
    m = DOFASegmentationModel(encoder="dofa_base", image_size=(512, 512), freeze_layers=["encoder"],
                              num_classes=5, pretrained=False)
    ref = oracle.DOFASegmentationModel("dofa_base", (512, 512), num_classes=5)
    sd, rsd = m.state_dict(), ref.state_dict()
    assert len(sd) == 332 and list(sd) == list(rsd)
    for k in sd:
        assert sd[k].shape == rsd[k].shape, k
    assert sum(p.numel() for p in m.parameters()) == 141_254_922
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 35_023_882
    # conv weights are stored channels-last but load/save with the logical OIHW shape
    w = m.neck.convs[0].conv.weight
    assert w.shape == (768, 768, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()
    m.load_state_dict(oracle.procedural_state_dict(ref, 1))
    assert m.neck.convs[0].conv.weight.permute(0, 2, 3, 1).is_contiguous()
    assert torch.equal(m.state_dict()["neck.convs.0.conv.weight"], oracle.procedural_state_dict(ref, 1)["neck.convs.0.conv.weight"])
    assert SegmentationOutput._fields == ("out", "aux")
    with pytest.raises(ValueError):
        DOFASegmentationModel(encoder="nope", pretrained=False)
    with pytest.raises(RuntimeError):
        DOFASegmentationModel(encoder="dofa_base", pretrained=True)  # no network: fails loudly


def test_neck_argument_errors():
    from geo_deep_learning.models.necks.multilevel_neck import MultiLevelNeck
    with pytest.raises(TypeError):
        MultiLevelNeck(64, [64], scales=[1], norm_cfg={"type": "BN"}, act_cfg={"type": "ReLU"})
    neck = MultiLevelNeck([64] * 2, [64] * 2, scales=[2, 1], norm_cfg={"type": "BN"}, act_cfg={"type": "ReLU"})
    with pytest.raises(ValueError, match="len\\(inputs\\)"):
        neck([torch.zeros(1, 64, 4, 4)])


def test_pos_embed_matches_reference_init():
    from geo_deep_learning.models.encoders.dofa_v2 import DOFAv2
    from oracle.encoder import get_2d_sincos_pos_embed
    e = DOFAv2(img_size=56, embed_dim=64, depth=1, num_heads=1, pretrained=False)
    assert torch.equal(e.pos_embed[0], get_2d_sincos_pos_embed(64, 4, cls_token=True))
    assert not e.pos_embed.requires_grad


def test_no_kernel_spills_to_scratch():
    """The build keeps hipcc's per-kernel resource remarks (csrc/build/*.usage): no kernel of the library may
    use scratch memory (a spilling MFMA kernel once cost 3x end to end), and the production conv / weight-
    gradient / attention kernels must keep >= 2 waves per SIMD."""
    import re
    from pathlib import Path
    build = Path(__file__).resolve().parents[1] / "geo-deep-learning_amd" / "csrc" / "build"
    files = sorted(build.glob("*.usage"))
    if not files:
        pytest.skip("no build logs (run __graft_entry__.build() first)")
    seen = 0
    for f in files:
        name = None
        for line in f.read_text().splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                seen += 1
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m:
                assert int(m.group(1)) == 0, f"{name} uses {m.group(1)} bytes/lane of scratch"
            m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
            if m and name and any(k in name for k in ("conv_gemm_kernel", "wgrad_tr_kernel", "flash_fwd_kernel")):
                assert int(m.group(1)) >= 2, f"{name}: occupancy {m.group(1)}"
    assert seen > 100


def test_packed_weight_cache_rejects_recycled_ids():
    """The packed-operand cache is keyed by id(): an entry built from a dead tensor whose id / address were
    recycled must not be served for the new tensor."""
    from gdlhip import nn as gnn
    p1 = torch.nn.Parameter(torch.ones(4))
    assert gnn.cached((p1,), "t_cache", lambda: "first") == "first"
    assert gnn.cached((p1,), "t_cache", lambda: "again") == "first"          # same object, same version: hit
    p2 = torch.nn.Parameter(torch.ones(4))
    key1, key2 = ("t_cache", id(p1)), ("t_cache", id(p2))
    ver, val, refs = gnn._CACHE[key1]
    gnn._CACHE[key2] = ((ver[0][:2] + (p2.data_ptr(),),), val, refs)         # what a recycled id would find
    assert gnn.cached((p2,), "t_cache", lambda: "second") == "second"
    with torch.no_grad():
        p2.add_(1)
    assert gnn.cached((p2,), "t_cache", lambda: "third") == "third"          # in-place update bumps the version


# ------------------------------------------------------------------ host-side kernel selection (pure host code)
def _conv_args(B, H, W, C, N, R, dtype=1, stride=1, aux=False, act=0):
    from gdlhip._lib import ConvArgs
    a = ConvArgs()
    a.inp, a.w, a.out = 0x1000, 0x2000, 0x3000
    a.dtype = a.out_dtype = dtype
    a.B, a.H, a.W, a.C = B, H, W, C
    a.in_sW, a.in_sH, a.in_sB = C, W * C, H * W * C
    pad = R // 2
    a.Ho, a.Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    a.R, a.S, a.stride, a.pad, a.N = R, R, stride, pad, N
    a.w_sN = R * R * C
    a.out_sW, a.out_sH, a.out_sB = N, a.Wo * N, a.Ho * a.Wo * N
    a.alpha, a.nz, a.nz_inner = 1.0, 1, 1
    a.act = act
    if aux:
        a.aux_out = 0x4000
    return a


def test_conv_tile_selection(lib):
    """gdl_conv_gemm_plan: which tile a layer gets (0 = 64^2, 1 = 128^2, 3 = 256^2 ping-pong, 4 = 3x3 shared staging,
    5 = 256x64, 6 = dual-resident 256x128, 7 = direct 3x3 for C <= 32 on large dense maps, 8 = 256^2 with one wave per SIMD,
    9 = persistent 256^2 ping-pong, 10 = persistent parked tile) and the algorithmic flops it reports."""
    from gdlhip import ops
    import ctypes as C

    def plan(*a, **k):
        fl = C.c_int64()
        v = lib.gdl_conv_gemm_plan(C.byref(_conv_args(*a, **k)), C.byref(fl))
        return v, fl.value
    assert plan(32, 1, 1297, 768, 768, 1)[0] == 9            # ViT proj: 163 x 3 = 489 tiles take the 256^2 tile (round 3), persistent (round 4)
    assert plan(4, 1, 1297, 768, 768, 1)[0] in (0, 1)        # the same layer at the reference's batch of 4: small tiles
    v, fl = plan(32, 144, 144, 768, 768, 3)                  # DOFA neck conv: 108 K-steps -> one wave per SIMD (round 4)
    assert v == 8 and fl == 2 * 32 * 144 * 144 * 768 * 9 * 768
    assert plan(32, 144, 144, 768, 256, 1)[0] == 9           # lateral 1x1: persistent 256^2 ping-pong (dense 1x1, more tiles than CUs)
    assert plan(1, 1, 32 * 1297, 768, 2304, 1)[0] == 9       # ViT qkv
    # the parked tile (variant 10, round 6) is opt-in: faster per layer, +-0 end to end (see gdl_conv_gemm_plan)
    lib.gdl_debug_set_conv_w4p.argtypes = [C.c_int]
    lib.gdl_debug_set_conv_w4p(1)
    try:
        assert plan(1, 1, 32 * 1297, 768, 2304, 1)[0] == 10     # N >= 768, 12 K-steps, 1467 tiles
        assert plan(32, 36, 36, 768, 6912, 1)[0] == 10          # the neck's nine tap products in one GEMM
        assert plan(32, 72, 72, 256, 2304, 1)[0] == 9           # K = 256 (4 K-steps): HBM-bound, stays on the 8-wave persistent tile
        assert plan(32, 144, 144, 768, 256, 1)[0] == 9          # N = 256: one tile per row block, memory-bound
    finally:
        lib.gdl_debug_set_conv_w4p(0)
    assert plan(1, 1, 32 * 1297, 768, 3072, 1, act=ops.ACT_GELU)[0] == 6   # ViT fc1: GELU epilogue, 12 K-steps -> two workgroups per CU
    assert plan(32, 1, 1297, 3072, 768, 1)[0] == 8           # ViT fc2: 48 K-steps -> one wave per SIMD
    assert plan(32, 36, 36, 768, 768, 3)[0] == 8             # neck 3x3 at 36^2: 108 K-steps
    assert plan(32, 144, 144, 256, 256, 3)[0] == 4           # FPN 3x3: 36 K-steps stay on the shared-staging kernel
    assert plan(8, 36, 36, 768, 768, 1)[0] in (0, 1, 3)      # 41 x 3 tiles: fewer tiles than CUs, nothing to be persistent about
    assert plan(32, 256, 256, 64, 64, 3)[0] == 5             # UNet++ decoder: narrow output on a large map
    assert plan(32, 512, 512, 32, 16, 3)[0] == 7             # 32 -> 16 channels at 512^2: direct kernel, one staged window
    assert plan(32, 256, 256, 32, 320, 3)[0] == 7            # its data gradient's shape: outputs in 32-channel slices
    assert plan(2, 36, 36, 32, 16, 3)[0] != 7                # map width not a multiple of the 64-pixel window
    assert plan(2, 16, 16, 64, 64, 3)[0] == 0                # too few pixels for 256-row tiles
    assert plan(32, 128, 128, 160, 256, 1)[0] in (0, 1)      # channel tail (MiT-B0 width): small tiles only
    assert plan(32, 128, 128, 768, 3072, 1, aux=True)[0] in (0, 1)   # training epilogue: never the 256^2 tiles


def test_conv_stats_rows_selection(lib):
    """gdl_conv_gemm_stats_rows (pure host code): which calls can emit BatchNorm partial statistics from their epilogue, and how
    many partial rows (one per 32 * TM output pixels: 128 on the 256^2 kernels, 64 on the 128^2 kernel)."""
    import ctypes as C

    def rows(*a, bias=0x5000, **k):
        args = _conv_args(*a, **k)
        args.bias = bias
        return lib.gdl_conv_gemm_stats_rows(C.byref(args))
    m144 = 32 * 144 * 144
    assert rows(32, 144, 144, 768, 256, 1) == m144 // 128        # UperNet lateral: persistent 256^2 tile
    assert rows(32, 144, 144, 256, 256, 3) == m144 // 128        # FPN 3x3: shared-staging kernel
    assert rows(32, 36, 36, 768, 768, 3) == 32 * 36 * 36 // 128  # neck 3x3: one wave per SIMD
    assert rows(32, 36, 36, 768, 256, 1) == 32 * 36 * 36 // 64   # 162 tiles of 256^2 are too few: 128^2 tile, 64 pixels per row
    assert rows(32, 144, 144, 768, 256, 1, dtype=0) == 0         # f32 parity path
    assert rows(32, 144, 144, 768, 200, 1) == 0                  # N tail
    assert rows(1, 25, 25, 256, 256, 1) == 0                     # no whole tiles
    assert rows(32, 6, 6, 768, 256, 1) == 0                      # pyramid-pooling branch: 64^2 tiles, register epilogue
    assert rows(32, 144, 144, 768, 256, 1, act=1) == 0           # anything but a bias-only epilogue
    a = _conv_args(32, 144, 144, 768, 256, 1)
    a.bias, a.resid = 0x5000, 0x6000
    assert lib.gdl_conv_gemm_stats_rows(C.byref(a)) == 0          # residual operand (fpn_bottleneck's native level)


def test_wgrad_split_selection(lib):
    """gdl_conv_wgrad_workspace = splits * N * R*S*C * 4: the row-segment kernel aims at two blocks per CU."""
    import ctypes as C
    from gdlhip._lib import WgradArgs

    def splits(B, H, W, Cc, N, R=3, nz=1):
        a = WgradArgs()
        a.inp, a.dy, a.dw, a.dtype = 0x1000, 0x2000, 0x3000, 1
        a.B, a.H, a.W, a.C = B, H, W, Cc
        a.in_sW, a.in_sH, a.in_sB = Cc, W * Cc, H * W * Cc
        a.Ho, a.Wo, a.R, a.S, a.stride, a.pad, a.N = H, W, R, R, 1, R // 2, N
        a.dy_sW, a.dy_sH, a.dy_sB = N, W * N, H * W * N
        a.dw_sN, a.nz, a.nz_inner = R * R * Cc, nz, 1
        return lib.gdl_conv_wgrad_workspace(C.byref(a)) // (max(nz, 1) * N * R * R * Cc * 4)
    # round 4: the 256^2 per-tap / 1x1 kernel holds a CU alone and its split count comes from a cost model over rounds of 256
    # workgroups (the old rule always ended in a nearly empty third round: 9 tiles x 57 splits = 513 workgroups)
    def fill(tiles, sp):
        blocks = tiles * sp
        return blocks / (-(-blocks // 256) * 256)
    assert fill(81, splits(32, 144, 144, 768, 768)) >= 0.9      # widest 3x3 layers: per-tap kernel, 3 x 9 x 3 tiles
    sp = splits(32, 36, 36, 768, 768, R=1)                      # neck lateral: 9 tiles, ONE round of at most 256 workgroups
    assert 9 * sp <= 256 and fill(9, sp) >= 0.85
    assert splits(32, 36, 36, 768, 6912, R=1) == 3              # tap weight gradient: 81 tiles x 3 = 243
    assert fill(3, splits(32, 144, 144, 768, 256, R=1)) >= 0.9 and splits(32, 144, 144, 768, 256, R=1) < 171
    assert splits(32, 144, 144, 256, 256) == 32
    assert splits(32, 256, 256, 64, 64) == 512               # one tile: all the parallelism comes from split-K
    assert splits(2, 16, 16, 256, 256) <= 1                  # narrow map: per-tap kernel, too few pixels to split
    assert splits(1, 1, 524288, 64, 256, R=1) > 64           # narrow linear over many pixels
    assert splits(4, 1, 1297, 64, 1297, R=1, nz=48) <= 64    # batched (attention dK / dV): nz-fold parallelism


def test_dynamic_segformer_mirror_has_the_reference_state_dict():
    """use_dynamic_encoder=True (segmentation_segformer.py:47): the mirror's parameter names and shapes are those of the
    oracle restatement, which is pinned to the real reference (tests/golden/segformer_dynamic.npz); no patch_embed1."""
    from geo_deep_learning.models.segmentation.segformer import SegFormerSegmentationModel
    from oracle.segformer import SegFormerSegmentationModel as OracleSegFormer
    for enc in ("mit_b0", "mit_b2"):
        m = SegFormerSegmentationModel(enc, 3, None, None, 5, use_dynamic_encoder=True)
        o = OracleSegFormer(enc, 3, 5, use_dynamic_encoder=True)
        ms, os_ = m.state_dict(), o.state_dict()
        assert sorted(ms) == sorted(os_)
        assert all(tuple(ms[k].shape) == tuple(os_[k].shape) for k in ms)
        assert not any(k.startswith("encoder.patch_embed1.") for k in ms)
        assert any(k.startswith("encoder.dynamic_patch_embed1.channel_attention.") for k in ms)
    with pytest.raises(ValueError):                       # no CPU fallback: the stem needs the device library
        SegFormerSegmentationModel("mit_b0", 3, None, None, 5, use_dynamic_encoder=True)(torch.zeros(1, 4, 32, 32))


def test_counter_batch_defers_batchnorm_counters_to_one_update():
    """gnn.counter_batch (host logic, CPU): inside the context `bump` only collects the BatchNorm step counters and the exit
    applies them all at once -- also when the body raises -- while outside of it (and in a nested context) the increment is
    what nn.BatchNorm2d.forward does: immediate, once."""
    import torch
    from gdlhip import nn as gnn
    counters = [torch.zeros((), dtype=torch.long) for _ in range(5)]
    gnn.bump(counters[0])
    assert int(counters[0]) == 1                       # outside: at once
    gnn.bump(None)                                     # track_running_stats=False modules have no counter
    with gnn.counter_batch():
        for c in counters:
            gnn.bump(c)
        with gnn.counter_batch():                      # nested: the outermost context applies
            gnn.bump(counters[1])
        assert [int(c) for c in counters] == [1, 0, 0, 0, 0]
    assert [int(c) for c in counters] == [2, 2, 1, 1, 1]
    try:
        with gnn.counter_batch():
            gnn.bump(counters[4])
            raise RuntimeError("forward failed after this layer ran")
    except RuntimeError:
        pass
    assert int(counters[4]) == 2 and gnn._COUNTER_BATCH is None


def test_drop_path_scales_semantics_on_cpu():
    """gnn.drop_path_scales (pure torch, runs on the CPU too): timm's drop_path with scale_by_keep -- a kept sample is scaled
    by 1 / keep, a dropped one by 0 -- drawn for every block of an encoder pass at once; blocks that never drop get None."""
    import torch
    from gdlhip import nn as gnn
    torch.manual_seed(0)
    probs = [0.0, 0.1, 0.0, 0.3]
    out = gnn.drop_path_scales(probs, 50000, torch.device("cpu"))
    assert out[0] == (None, None) and out[2] == (None, None)
    for p_, pair in ((0.1, out[1]), (0.3, out[3])):
        for t in pair:
            vals = sorted(t.unique().tolist())
            assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p_)) < 1e-6
            assert abs(float((t > 0).float().mean()) - (1.0 - p_)) < 0.01
            assert abs(float(t.mean()) - 1.0) < 0.02          # unbiased: E[scale] = 1
    assert gnn.drop_path_scales([0.0, 0.0], 4, torch.device("cpu")) == [(None, None), (None, None)]


def test_unetpp_bottleneck_mirror_keys_and_cached_imagenet_weights(tmp_path, monkeypatch):
    """The UNet++ mirror accepts the encoder of the reference's shipped config (resnext101_32x8d,
    configs/unetplus_config_RGB.yaml:37-39): same state-dict keys and shapes as the oracle (= torchvision / smp names), and
    ``encoder_weights="imagenet"`` reads torchvision's checkpoint from the torch-hub cache (never downloads): strict load
    with ``fc.*`` dropped, a clear error that names the path when the file is not there."""
    import torch
    from geo_deep_learning.models.segmentation.unetplusplus import UnetPlusPlus
    from oracle.unetpp import ResNetEncoder as OracleEncoder
    from oracle.unetpp import UnetPlusPlus as OracleUnetPlusPlus
    for name in ("resnext50_32x4d", "resnet50"):
        ora, m = OracleUnetPlusPlus(name, 3, 2), UnetPlusPlus(name, encoder_weights=None, classes=2)
        want = {k: tuple(v.shape) for k, v in ora.state_dict().items()}
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want, name
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path))
    with pytest.raises(RuntimeError, match="checkpoints/resnext50_32x4d"):
        UnetPlusPlus("resnext50_32x4d", encoder_weights="imagenet", classes=2)
    src = OracleEncoder("resnext50_32x4d", 3)
    sd = {k: torch.randn_like(v) if v.dtype.is_floating_point else v for k, v in src.state_dict().items()}
    sd["fc.weight"], sd["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)      # the classification head smp drops
    (tmp_path / "checkpoints").mkdir()
    torch.save(sd, tmp_path / "checkpoints" / "resnext50_32x4d-7cdf4587.pth")
    m = UnetPlusPlus("resnext50_32x4d", encoder_weights="imagenet", classes=2)
    for k, v in m.encoder.state_dict().items():
        assert torch.equal(v, sd[k]), k
    with pytest.raises(NotImplementedError, match="not built"):
        UnetPlusPlus("efficientnet-b0", encoder_weights=None)


def test_bench_stdout_line_is_short_and_ends_with_the_headline_extras():
    """bench.py prints ONE stdout line that a driver keeping only a 2000-character tail still reads in full where it matters
    (round-4 review, evidence hygiene): under ~3.2 KB for a complete result, contract keys first, and the inference rate, both
    utilisations, the sustained run, step-level roofline fractions, small batches and the other BASELINE configs inside the tail.
    Input: a committed full line of round 4 (data), extended by the round-5 `sustained` entry."""
    import importlib.util
    import json
    root = Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("bench_for_test", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    full = json.loads((root / "profiles" / "r04w_bench_dofa_b32_full_line_final_code.json").read_text().strip().splitlines()[-1])
    full["sustained"] = {"train_tiles_per_s": 880.1, "inference_tiles_per_s": 1700.2, "steps": {"train": 90, "infer": 160},
                         "timed_region_s": {"train": 3.1, "infer": 3.0}, "note": "x" * 200}
    line = bench.compact_line(full, ["bench_details.json"])
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= 3300, len(text)
    keys = list(line)
    assert keys[:13] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data", "config"]
    assert keys[-1] == "inference_tiles_per_s" and line["value"] == full["value"]
    tail = text[-2000:]
    for needle in ('"inference_tiles_per_s"', '"model_flops_utilisation"', '"executed_flops_utilisation"', '"sustained"',
                   '"step_roofline"', '"by_batch"', '"dofa_large_1024_10band"', '"pcie_inclusive"'):
        assert needle in tail, needle
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])


def test_dofa_training_step_asks_for_low_resolution_logits_only_where_the_loss_can_use_them(monkeypatch):
    """Host logic of tasks_with_models/segmentation_dofa.py: ``training_step`` (round 5) and ``validation_step`` / ``test_step``
    (round 6: the mask comes from ``gnn.predict_mask`` on the same low-resolution map) hand the heads' not-yet-resized maps
    (``lowres_logits=True``) to gdlhip's multiclass DiceLoss and to nothing else; a binary or foreign loss, one class, and
    GDL_LOWRES_DICE=0 get the reference's full-resolution logits (dofa.py:89-105)."""
    from types import SimpleNamespace
    from gdlhip import nn as gnn
    from tasks_with_models.segmentation_dofa import SegmentationDOFA
    calls = []

    class Model(torch.nn.Module):
        def forward(self, x, wv, lowres_logits=False):
            calls.append(bool(lowres_logits))
            out = torch.zeros(x.shape[0], 5, 8, 8, requires_grad=True)
            return SimpleNamespace(out=out, aux=out)

    class FakeDice(gnn.DiceLoss):       # the class the task tests for; no kernel behind it here
        def forward(self, y_pred, y_true):
            return y_pred.sum() * 0.0

    class Foreign(torch.nn.Module):
        def forward(self, y_pred, y_true):
            return y_pred.sum() * 0.0

    def task_with(loss):
        t = SegmentationDOFA("dofa_base", pretrained=False, image_size=(8, 8), num_classes=5, max_samples=1, loss=loss)
        t.model = Model()
        return t

    batch = {"image": torch.zeros(2, 3, 8, 8), "mask": torch.zeros(2, 1, 8, 8, dtype=torch.int64), "wavelengths": torch.tensor([0.6, 0.5, 0.4])}
    monkeypatch.setattr(gnn, "FUSE_LOWRES_DICE", True)
    t = task_with(FakeDice(mode="multiclass"))
    t.training_step(batch, 0)
    assert calls == [True]
    calls.clear()
    seen = []
    monkeypatch.setattr(gnn, "predict_mask", lambda logits: (seen.append(logits), logits.argmax(1))[1])      # (the mask kernel needs a GPU)
    with torch.no_grad():
        t.validation_step(batch, 0)
    assert calls == [True] and len(seen) == 1, "validation: loss and mask both come from the low-resolution maps"
    calls.clear()
    for other in (task_with(FakeDice(mode="binary")), task_with(Foreign())):
        with torch.no_grad():
            other.validation_step(batch, 0)
    assert calls == [False, False], "a loss that cannot read low-resolution maps gets the resized logits in validation too"
    calls.clear()
    task_with(FakeDice(mode="binary")).training_step(batch, 0)
    task_with(Foreign()).training_step(batch, 0)
    assert calls == [False, False]
    calls.clear()
    monkeypatch.setattr(gnn, "FUSE_LOWRES_DICE", False)
    t = task_with(FakeDice(mode="multiclass"))
    t.training_step(batch, 0)
    with torch.no_grad():
        t.validation_step(batch, 0)
    assert calls == [False, False]
