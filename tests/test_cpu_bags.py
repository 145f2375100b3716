"""stamp_amd.bags (patients -> bags -> batches -> loss weights) against what the reference's own `BagDataset.__getitem__`, `_collate_to_tuple`,
`_collate_multitarget`, `_parse_targets` and `_compute_class_weights_and_check_categories` returned (tests/golden/bag_dataset.json, made by
tools/make_golden.py::golden_bag_dataset from /root/reference/src/stamp/modeling/data.py:85-295, 532-655, 811-862 and modeling/train.py:567-621) --
here over REAL feature files written by stamp_amd.h5io."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from stamp_amd import bags as B
from stamp_amd import h5io

G = json.loads((Path(__file__).parent / "golden" / "bag_dataset.json").read_text())


@pytest.fixture(scope="module")
def feature_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("feats")
    for name, rec in G["files"].items():
        ds = {k: np.asarray(v["data"], dtype=v["dtype"]) for k, v in rec["datasets"].items()}
        h5io._write(d / f"{name}.h5", ds, rec["attrs"])
    return d


def _bags(feature_dir):
    return [[feature_dir / f"{n}.h5" for n in b] for b in G["bags"]]


def _nan_eq(a, b):
    a = torch.as_tensor([[math.nan if v is None else v for v in r] for r in a], dtype=torch.float32)
    return a.shape == b.shape and bool(((a == b) | (a.isnan() & b.isnan())).all())


def test_parse_targets_equals_the_reference():
    t = G["targets"]
    pd = lambda gts: [B.PatientData(ground_truth=(tuple(g) if isinstance(g, list) else g), feature_files=[]) for g in gts]  # noqa: E731
    y, cats = B.parse_targets(patient_data=pd(t["classification"]["gts"]), task="classification")
    assert cats == t["classification"]["categories"] and y.dtype == torch.float32 and y.tolist() == t["classification"]["encoded"]
    y, cats = B.parse_targets(patient_data=pd(t["classification"]["gts"]), task="classification", categories=t["classification_fixed"]["categories_in"])
    assert cats == t["classification_fixed"]["categories"] and y.tolist() == t["classification_fixed"]["encoded"]
    y, cats = B.parse_targets(patient_data=pd(t["regression"]["gts"]), task="regression")
    assert cats == [] and _nan_eq(t["regression"]["encoded"], y)
    y, _ = B.parse_targets(patient_data=pd(t["survival"]["gts"]), task="survival")
    assert _nan_eq(t["survival"]["encoded"], y)
    y, cats = B.parse_targets(patient_data=pd(t["multi"]["gts"]), task="classification")
    assert cats == t["multi"]["categories"] and [{k: v.tolist() for k, v in d.items()} for d in y] == t["multi"]["encoded"]
    with pytest.raises(ValueError) as e:
        B.parse_targets(patient_data=pd(["x"] * 5), task="classification")
    assert str(e.value) == t["one_class_error"]
    with pytest.raises(ValueError):
        B.parse_targets(patient_data=pd([1.0]), task="ranking")
    with pytest.raises(ValueError):
        B.parse_targets(patient_data=pd([5.0]), task="survival")


def _check_item(it, rec):
    bag, coords, n, tgt = it
    assert str(bag.dtype) == rec["bag_dtype"] == "torch.float32" and coords.dtype == torch.float32
    assert torch.equal(bag, torch.tensor(rec["bag"], dtype=torch.float32).reshape(bag.shape)), "bag rows differ"
    assert torch.equal(coords, torch.tensor(rec["coords"], dtype=torch.float32).reshape(coords.shape)) and int(n) == rec["n"]
    if isinstance(tgt, dict):
        assert {k: v.tolist() for k, v in tgt.items()} == rec["target"]
    else:
        assert torch.as_tensor(tgt).tolist() == rec["target"]


@pytest.mark.parametrize("case", ["det_16", "rand_16", "rand_40", "all"])
def test_bag_dataset_items_and_batches_equal_the_reference(feature_dir, case):
    """multi-slide patients concatenated in file order, fp16 files -> `.float()`, three coordinate conventions, the coords-less bypass;
    equidistant and seeded random sampling (the same torch.randperm draws), zero padding, bag_size=None."""
    rec = G["cases"][case]
    y = torch.tensor(G["targets"]["classification"]["encoded"])
    ds = B.BagDataset(bags=_bags(feature_dir), ground_truths=y, transform=None, bag_size=rec["kw"]["bag_size"], deterministic=rec["kw"].get("deterministic", False))
    if rec["seed"] is not None:
        torch.manual_seed(rec["seed"])
    items = [ds[i] for i in range(len(ds))]
    for it, r in zip(items, rec["items"]):
        _check_item(it, r)
    if "batch" in rec:
        b, c, s, t = B.collate_to_tuple(items)
        assert list(b.shape) == rec["batch"]["bags_shape"] and list(c.shape) == rec["batch"]["coords_shape"]
        assert s.tolist() == rec["batch"]["bag_sizes"] and str(s.dtype) == rec["batch"]["bag_sizes_dtype"] and t.tolist() == rec["batch"]["targets"]


def test_transform_multitarget_and_collate_shape_rule(feature_dir):
    rec = G["cases"]["multi_det_8_transform"]
    y, _ = B.parse_targets(patient_data=[B.PatientData(ground_truth=g, feature_files=[]) for g in G["targets"]["multi"]["gts"]], task="classification")
    ds = B.BagDataset(bags=_bags(feature_dir), bag_size=8, ground_truths=y, transform=lambda x: x * 2.0 + 1.0, deterministic=True)
    items = [ds[i] for i in range(len(ds))]
    for it, r in zip(items, rec["items"]):
        _check_item(it, r)
    b, _, s, t = B.collate_multitarget(items)
    assert list(b.shape) == rec["batch"]["bags_shape"] and s.tolist() == rec["batch"]["bag_sizes"] and {k: v.tolist() for k, v in t.items()} == rec["batch"]["targets"]
    cs = G["cases"]["collate_shapes"]
    fake = [(items[0][0], items[0][1], items[0][2], torch.tensor(cs["targets_in"][0])), (items[1][0], items[1][1], items[1][2], torch.tensor(cs["targets_in"][1]))]
    assert B.collate_to_tuple(fake)[3].tolist() == cs["targets_out"]
    with pytest.raises(ValueError, match="number of ground truths"):
        B.BagDataset(bags=_bags(feature_dir), ground_truths=torch.zeros(2, 2), transform=None)


def test_open_file_limit_and_eviction_order(feature_dir, tmp_path):
    rec = G["cases"]["handle_cache"]
    src = G["files"]["a2"]
    ds_arrays = {k: np.asarray(v["data"], dtype=v["dtype"]) for k, v in src["datasets"].items()}
    paths = []
    for i in range(140):
        p = tmp_path / f"many{i}.h5"
        h5io._write(p, ds_arrays, src["attrs"])
        paths.append([p])
    ds = B.BagDataset(bags=paths, bag_size=2, ground_truths=torch.zeros(140, 2), transform=None, deterministic=True)
    for i in list(range(140)) + [0, 139, 11]:
        ds[i]
    assert ds._files.opens == rec["opens"] and len(ds._files.entries) == rec["cached_after"] == B.MAX_OPEN_FILES
    import pickle
    clone = pickle.loads(pickle.dumps(ds))               # what a spawn-started DataLoader worker receives: no open files
    assert len(clone._files.entries) == 0 and torch.equal(clone[3][0], ds[3][0])


def test_class_weights_equal_the_reference(caplog):
    cw = G["class_weights"]
    gt = torch.tensor(cw["single"]["ground_truths"])
    with caplog.at_level("WARNING", logger="stamp"):
        w = B.class_weights(gt, ["a", "b", "c"])
    assert w.tolist() == cw["single"]["weights"] and abs(float(w.sum()) - 1.0) < 1e-6 and "{'c': 10}" in caplog.text
    y, cats = B.parse_targets(patient_data=[B.PatientData(ground_truth=g, feature_files=[]) for g in G["targets"]["multi"]["gts"]], task="classification")
    wm = B.class_weights(y, cats)
    assert {k: [repr(float(x)) for x in v.tolist()] for k, v in wm.items()} == cw["multi_raw"]
    with pytest.raises(ValueError) as e:
        B.class_weights(gt, ["only"])
    assert str(e.value) == cw["one_category_error"]


def test_dataloader_feeds_the_reference_batch_contract(feature_dir):
    pdata = [B.PatientData(ground_truth=g, feature_files=b) for g, b in zip(G["targets"]["classification"]["gts"], _bags(feature_dir))]
    dl, cats = B.tile_bag_dataloader(patient_data=pdata, bag_size=16, task="classification", batch_size=2, shuffle=False, num_workers=0, transform=None)
    assert cats == G["targets"]["classification"]["categories"]
    batches = list(dl)
    assert [tuple(b[0].shape) for b in batches] == [(2, 16, 12), (2, 16, 12), (1, 16, 12)] and batches[0][1].shape == (2, 16, 2)
    rec = G["cases"]["det_16"]                              # shuffle=False -> deterministic sampling (data.py:121)
    assert torch.equal(batches[0][0][1], torch.tensor(rec["items"][1]["bag"])) and batches[2][2].tolist() == [rec["items"][4]["n"]]
