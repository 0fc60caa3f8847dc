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
    proc = SampleProcessor("sensorA", norm, "dofa")
    r = proc(_sample(g, "u8"))
    assert r["image"].dtype == torch.uint8 and r["image"].shape == (4, 24, 24)
    b = collate([r, proc(_sample(g, "u8"))])
    assert b["image"].shape == (2, 4, 24, 24) and b["mean"].shape == (2, 4, 1, 1) and b["image_name"] == ["k0", "k0"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_device_input_stage_matches_reference(stats, kind):
    from geo_deep_learning.datamodules.device_input import DeviceInputStage
    g, norm = stats
    proc = SampleProcessor("sensorA", norm, "dofa")
    batches = []
    for i in range(7):    # more batches than ring slots: buffers are reused while copies are in flight
        s = _sample(g, kind)
        s["image_patch.npy"] = np.roll(s["image_patch.npy"], i, axis=2)
        batches.append(collate([proc(s), proc(s)]))
    stage = DeviceInputStage(batches, "cuda", depth=2)
    ref = torch.from_numpy(g[f"{kind}_dofa_image"])
    n = 0
    for i, b in enumerate(stage):
        assert b["image"].is_cuda and b["image"].dtype == torch.float32 and b["mask"].is_cuda
        assert not b["wavelengths"].is_cuda
        want = torch.roll(ref, i, dims=2)
        got = b["image"].cpu()
        assert torch.equal(got[0], want) and torch.equal(got[1], want), f"batch {i}"     # bit-exact vs reference
        np.testing.assert_array_equal(b["mask"][0].cpu().numpy(), g[f"lab_{kind}"].astype(np.int64))
        n += 1
    assert n == 7
    raw_bytes = batches[0]["image"].numel() * batches[0]["image"].element_size()
    assert stage.bytes_h2d < 7 * (raw_bytes + 2 * 24 * 24 * 8 + 4096)     # raw tiles, not f32, crossed PCIe
