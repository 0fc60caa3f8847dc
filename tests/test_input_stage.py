"""X2 (tile decode + normalise + H2D): the dataset-side sample processing against goldens produced by the
REAL reference's ShardedDataset._process_sample (CPU), and the GPU input stage (pinned ring + copy stream +
gdl_normalize_raw) against the same goldens."""

import json

import numpy as np
import pytest
import torch

from geo_deep_learning.datasets.wds_dataset import SampleProcessor, collate, load_normalization_stats

KINDS = ("u8", "u16", "i16")
LAYOUTS = ("clay", "dofa", "unified")


def _sample(g, kind):
    return {"__key__": "k0", "image_patch.npy": g[f"img_{kind}"], "label_patch.npy": g[f"lab_{kind}"],
            "metadata.json": json.loads(str(g["meta_json"]))}


@pytest.fixture()
def stats(golden_dir, tmp_path):
    g = np.load(golden_dir / "process_sample.npz")
    p = tmp_path / "stats.json"
    p.write_text(str(g["stats_json"]))
    return g, load_normalization_stats(str(p), "sensorA")


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("layout", LAYOUTS)
def test_process_sample_matches_reference(stats, kind, layout):
    g, norm = stats
    proc = SampleProcessor("sensorA", norm, layout, defer_normalization=False)
    r = proc(_sample(g, kind))
    assert sorted(r.keys()) == json.loads(str(g[f"{kind}_{layout}_keys"]))
    assert r["image"].dtype == torch.float32 and r["mask"].dtype == torch.int64
    np.testing.assert_array_equal(r["image"].numpy(), g[f"{kind}_{layout}_image"])      # bit-exact host arithmetic
    np.testing.assert_array_equal(r["mask"].numpy(), g[f"{kind}_{layout}_mask"])
    for k in ("time", "latlon", "wavelengths", "mean", "std"):
        if f"{kind}_{layout}_{k}" in g.files:
            np.testing.assert_array_equal(r[k].numpy(), g[f"{kind}_{layout}_{k}"], err_msg=k)


def test_deferred_sample_keeps_raw_dtype(stats):
    g, norm = stats
    proc = SampleProcessor("sensorA", norm, "dofa", defer_normalization=True)
    r = proc(_sample(g, "u8"))
    assert r["image"].dtype == torch.uint8 and r["image"].shape == (4, 24, 24)
    b = collate([r, proc(_sample(g, "u8"))])
    assert b["image"].shape == (2, 4, 24, 24) and b["mean"].shape == (2, 4, 1, 1) and b["image_name"] == ["k0", "k0"]


@pytest.mark.gpu
@pytest.mark.parametrize("threaded", [True, False], ids=["worker-thread", "consumer-thread"])
@pytest.mark.parametrize("kind", KINDS)
def test_device_input_stage_matches_reference(stats, kind, threaded):
    """(round 5: the host side of the staging on a worker thread -- the default -- and on the consumer thread, as before)"""
    from geo_deep_learning.datamodules.device_input import DeviceInputStage
    g, norm = stats
    proc = SampleProcessor("sensorA", norm, "dofa", defer_normalization=True)
    batches = []
    for i in range(7):    # more batches than ring slots: buffers are reused while copies are in flight
        s = _sample(g, kind)
        s["image_patch.npy"] = np.roll(s["image_patch.npy"], i, axis=2)
        batches.append(collate([proc(s), proc(s)]))
    stage = DeviceInputStage(batches, "cuda", depth=2, threaded=threaded)
    ref = torch.from_numpy(g[f"{kind}_dofa_image"])
    n = 0
    for i, b in enumerate(stage):
        assert b["image"].is_cuda and b["image"].dtype == torch.float32 and b["mask"].is_cuda
        assert b["mask"].dtype == torch.int64            # crossed the link as uint8 (class indices), widened on the device
        assert not b["wavelengths"].is_cuda
        want = torch.roll(ref, i, dims=2)
        got = b["image"].cpu()
        assert torch.equal(got[0], want) and torch.equal(got[1], want), f"batch {i}"     # bit-exact vs reference
        np.testing.assert_array_equal(b["mask"][0].cpu().numpy(), g[f"lab_{kind}"].astype(np.int64))
        n += 1
    assert n == 7
    raw_bytes = batches[0]["image"].numel() * batches[0]["image"].element_size()
    assert stage.bytes_h2d < 7 * (raw_bytes + 2 * 24 * 24 * 1 + 4096)     # raw tiles, not f32, and byte masks crossed PCIe
    # a mask that does not fit a byte (e.g. an ignore index of 300) is shipped as it is
    wide = dict(batches[0])
    wide["mask"] = batches[0]["mask"].clone()
    wide["mask"].view(-1)[0] = 300
    (b,) = list(DeviceInputStage([wide], "cuda", depth=1, threaded=threaded))
    assert b["mask"].dtype == torch.int64 and int(b["mask"].view(-1)[0]) == 300
    # an exception in the source iterable reaches the consumer, a consumer that stops early does not hang the worker
    def broken():
        yield batches[0]
        raise KeyError("shard 7 is missing")
    with pytest.raises(KeyError, match="shard 7"):
        list(DeviceInputStage(broken(), "cuda", depth=2, threaded=threaded))
    for b in DeviceInputStage(batches, "cuda", depth=2, threaded=threaded):
        break


@pytest.mark.gpu
def test_augment_kernel_matches_torch():
    """Every augmentation kind vs torch.flip / torch.rot90 / F.interpolate on the crop (image bilinear, mask nearest),
    with and without the fused normalisation of a raw uint8 tile."""
    import torch.nn.functional as F
    from gdlhip import ops
    g = torch.Generator().manual_seed(0)
    B, C, H = 6, 3, 64
    u8 = torch.randint(0, 256, (B, C, H, H), generator=g, dtype=torch.uint8)
    mask = torch.randint(0, 5, (B, 1, H, H), generator=g, dtype=torch.int64)
    mean, std = torch.tensor([0.4, 0.43, 0.4]), torch.tensor([0.17, 0.18, 0.16])
    x = ((u8.float() / 255.0) - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    prm = torch.zeros(B, 8)
    prm[1, 0] = 1
    prm[2, 0] = 2
    prm[3, 0], prm[3, 1] = 3, 1
    prm[4, 0], prm[4, 1] = 3, 3
    prm[5, 0], prm[5, 2:6] = 4, torch.tensor([5.0, 9.0, 40.0, 31.0])
    want_img = [x[0], x[1].flip(-1), x[2].flip(-2), torch.rot90(x[3], 1, (-2, -1)), torch.rot90(x[4], 3, (-2, -1)),
                F.interpolate(x[5:6, :, 5:45, 9:40], size=(H, H), mode="bilinear", align_corners=False)[0]]
    want_mask = [mask[0], mask[1].flip(-1), mask[2].flip(-2), torch.rot90(mask[3], 1, (-2, -1)),
                 torch.rot90(mask[4], 3, (-2, -1)),
                 F.interpolate(mask[5:6, :, 5:45, 9:40].float(), size=(H, H), mode="nearest")[0].long()]
    for fused in (True, False):
        src = u8.to("cuda") if fused else x.to("cuda")
        img, om = ops.augment(src, mask.to("cuda").view(B, H, H), prm.to("cuda"),
                              mean.to("cuda") if fused else None, std.to("cuda") if fused else None)
        for i in range(B):
            assert torch.equal(om[i].cpu(), want_mask[i][0]), f"mask {i}"
            if i < 5:
                assert torch.equal(img[i].cpu(), want_img[i]), f"image {i} fused={fused}"     # pure index maps: bit-exact
            else:
                assert (img[i].cpu() - want_img[i]).abs().max().item() < 2e-6


@pytest.mark.gpu
def test_reference_pipeline_through_the_stage(stats):
    from gdlhip.augment import reference_pipeline
    from geo_deep_learning.datamodules.device_input import DeviceInputStage
    g, norm = stats
    proc = SampleProcessor("sensorA", norm, "dofa", defer_normalization=True)
    s = _sample(g, "u8")
    batches = [collate([proc(s)] * 4) for _ in range(12)]
    torch.manual_seed(5)
    stage = DeviceInputStage(batches, "cuda", depth=2, augment=reference_pipeline((24, 24)))
    ref = torch.from_numpy(g["u8_dofa_image"])
    changed = 0
    for b in stage:
        img, m = b["image"].cpu(), b["mask"].cpu()
        assert img.shape == (4, 4, 24, 24) and img.dtype == torch.float32 and m.dtype == torch.int64
        assert torch.isfinite(img).all() and int(m.min()) >= 0 and int(m.max()) <= 4
        # value range can only shrink under flips / turns / bilinear crops
        assert img.max() <= ref.max() + 1e-5 and img.min() >= ref.min() - 1e-5
        changed += int(any(not torch.equal(img[i], ref) for i in range(4)))
    assert changed >= 6      # p = 0.5 per sample, 4 samples, 12 batches
