"""The host layer between a directory of tile-feature `.h5` files and the MIL training step: patients -> bags -> batches.

Mirrors the reference's src/stamp/modeling/data.py (the part `stamp train` / `stamp crossval` run before the first model call):

    parse_targets          `_parse_targets` :146-252       raw ground truths -> one-hot / float / (time, event) tensors; the ONE place task semantics live
    BagDataset             `BagDataset` :532-655           patient -> its slides' feature files concatenated, `.float()`, transform, fixed-size bag
    collate_to_tuple       `_collate_to_tuple` :255-277    items -> (bags [B, bag, F], coords [B, bag, 2], bag_sizes [B], targets [B, D])
    collate_multitarget    `_collate_multitarget` :280-295 the same with a dict of target tensors
    tile_bag_dataloader    `tile_bag_dataloader` :85-143   the DataLoader `train_model_` consumes
    class_weights          `_compute_class_weights_and_check_categories`, modeling/train.py:567-621   inverse-frequency weights of the loss

Files are read through `stamp_amd.h5io` (h5py, else libhdf5 via ctypes, else the pure-Python subset), coordinates through its `get_coords`
(the reference's three conventions).  Sampling draws from torch's CPU generator exactly like the reference (`torch.randperm`), so a seeded run
picks the same tiles.  This is host code, as in the reference (DataLoader workers); with `device=` the row gather of the fixed-size bag runs on
the GPU (`amds_gather_rows`) for callers that keep a cohort's features resident in HBM.  Pinned by tests/golden/bag_dataset.json, made by
executing the reference's own definitions (tools/make_golden.py::golden_bag_dataset)."""
from __future__ import annotations

import logging
import math
from collections import OrderedDict
from collections.abc import Callable, Iterable, Mapping, Sequence
from dataclasses import KW_ONLY, dataclass, field
from pathlib import Path
from typing import Any

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import h5io

_log = logging.getLogger("stamp")
MAX_OPEN_FILES = 128          # data.py:598: "Limit open handles to avoid reaching OS ulimits"


@dataclass
class PatientData:
    """What the tables say about one patient (data.py:76-83): the raw ground truth and the feature files of the patient's slides."""
    _: KW_ONLY
    ground_truth: Any
    feature_files: Iterable


# ---- ground truths -> tensors -------------------------------------------------------------------------------------------------------------
def _nan_if_missing(v) -> float:
    return math.nan if v is None or str(v).lower() == "nan" else float(v)


def parse_targets(*, patient_data: Sequence[PatientData], task: str, categories: Sequence[str] | None = None):
    """-> (targets, categories).  classification: float one-hot [N, C] in the order of `categories` (default: the sorted set of labels seen; a
    patient with no label is all zeros), or -- when the ground truths are dicts -- a list of {target: one-hot} with per-target sorted categories;
    regression: float [N, 1] (missing = NaN); survival: float [N, 2] = (time, event), missing entries NaN."""
    gts = [p.ground_truth for p in patient_data]
    if task == "classification":
        dicts = [g for g in gts if isinstance(g, dict)]
        if dicts:
            names = list(dicts[0].keys())
            cats = {t: sorted({g[t] for g in dicts if g.get(t) is not None}) for t in names}
            enc = []
            for g in gts:
                row = {}
                for t in names:
                    v = g.get(t) if isinstance(g, dict) else None
                    row[t] = torch.zeros(len(cats[t])) if v is None else torch.tensor([v == c for c in cats[t]], dtype=torch.float32)
                enc.append(row)
            return enc, cats
        seen = {g for g in gts if g is not None}
        if len(seen) < 2 and categories is None:
            raise ValueError("Only one unique class found in classification task. This is usually a data or configuration error.")
        cats = list(categories) if categories else sorted(seen)
        return torch.tensor([[g == c for c in cats] for g in gts], dtype=torch.float32).reshape(len(gts), len(cats)), cats
    if task == "regression":
        return torch.tensor([math.nan if g is None else float(g) for g in gts], dtype=torch.float32).reshape(-1, 1), []
    if task == "survival":
        rows = []
        for g in gts:
            if g is None:
                rows.append((math.nan, math.nan))
            elif isinstance(g, (tuple, list)) and len(g) == 2:
                rows.append((_nan_if_missing(g[0]), math.nan if g[1] is None else float(g[1])))
            else:
                raise ValueError("survival ground truth must be a (time, event) tuple/list")
        return torch.tensor(np.asarray(rows, dtype=np.float64).reshape(-1, 2), dtype=torch.float32), []
    raise ValueError(f"Unsupported task: {task}")


# ---- bags -----------------------------------------------------------------------------------------------------------------------------------
def fixed_size_bag(feats: torch.Tensor, coords: torch.Tensor, bag_size: int, deterministic: bool = False, device=None):
    """data.py:811-862: at most `bag_size` rows -- all of them when the bag is small (zero rows appended), `bag_size` equidistant ones when
    `deterministic`, else the first `bag_size` of a random permutation -- and the number of real rows.  `device`: gather on the GPU."""
    from .mil import fixed_size_bag_indices, to_fixed_size_bag
    if device is not None:
        return to_fixed_size_bag(feats.to(device), coords.to(device), bag_size, deterministic)
    n = feats.shape[0]
    idx = fixed_size_bag_indices(n, bag_size, deterministic)
    bag, c = feats[idx], coords[idx]
    if bag.shape[0] < bag_size:
        bag = torch.cat((bag, bag.new_zeros(bag_size - bag.shape[0], bag.shape[1])))
        c = torch.cat((c, c.new_zeros(bag_size - c.shape[0], c.shape[1])))
    return bag, c, min(bag_size, n)


class _OpenFiles:
    """At most MAX_OPEN_FILES feature files open at a time, least recently used closed first (data.py:596-612).  With h5py a handle stays
    open between reads (the reference's behaviour); the ctypes / pure-Python backends of h5io keep nothing open, an entry is then only the
    bookkeeping that makes the eviction order the same."""

    def __init__(self) -> None:
        self.entries: OrderedDict = OrderedDict()
        self.opens = 0

    def read(self, path):
        if path in self.entries:
            self.entries.move_to_end(path)
        else:
            if len(self.entries) >= MAX_OPEN_FILES:
                _, h = self.entries.popitem(last=False)
                if hasattr(h, "close"):
                    h.close()
            self.entries[path] = self._open(path)
            self.opens += 1
        h = self.entries[path]
        if h is None:
            return h5io.read_file(path)
        return h5io.read_open_h5py(h)

    @staticmethod
    def _open(path):
        if h5io.backend() != "h5py":
            return None
        import h5py
        try:
            return h5py.File(path, "r", swmr=True, libver="latest")
        except Exception:          # older files / unconventional storage (data.py:608-610)
            return h5py.File(path, "r")


@dataclass
class BagDataset(Dataset):
    """One item per patient: (bag f32 [bag_size | n, F], coords f32 [.., 2] in micrometres, number of real rows, ground truth)."""
    _: KW_ONLY
    bags: Sequence[Iterable]
    """per patient: the `.h5` files of the patient's slides (each: `feats` [n, F] -- or `patch_embeddings` -- and, normally, `coords`)"""
    bag_size: int | None = None
    ground_truths: Any = None
    transform: Callable[[torch.Tensor], torch.Tensor] | None = None
    deterministic: bool = False
    tile_size_px: int | None = None
    device: Any = None
    _files: _OpenFiles = field(default_factory=_OpenFiles, init=False, repr=False, compare=False)

    def __post_init__(self) -> None:
        if self.ground_truths is None or len(self.bags) != len(self.ground_truths):
            raise ValueError("the number of ground truths has to match the number of bags")

    def __getstate__(self) -> dict:        # open files do not travel to DataLoader workers; each worker opens its own on first use
        state = self.__dict__.copy()
        state["_files"] = _OpenFiles()
        return state

    def __len__(self) -> int:
        return len(self.bags)

    def __getitem__(self, index: int):
        feats, coords = [], []
        for f in self.bags[index]:
            d, a = self._files.read(f)
            feats.append(torch.from_numpy(np.asarray(d["feats"] if "feats" in d else d["patch_embeddings"])))
            coords.append(torch.from_numpy(np.asarray(h5io.get_coords(d, a).coords_um)))
        bag, c = torch.concat(feats).float(), torch.concat(coords).float()
        if self.transform is not None:
            bag = self.transform(bag)
        if self.bag_size is None:
            return bag, c, len(bag), self.ground_truths[index]
        return (*fixed_size_bag(bag, c, self.bag_size, self.deterministic, self.device), self.ground_truths[index])


def _stack_items(items):
    return (torch.stack([it[0] for it in items]), torch.stack([it[1] for it in items]), torch.tensor([it[2] for it in items]))


def collate_to_tuple(items):
    """scalar targets become [1], anything with more than one axis is flattened, then stacked to [B, D] (data.py:264-277)"""
    bags, coords, sizes = _stack_items(items)
    tg = []
    for it in items:
        t = torch.as_tensor(it[3])
        tg.append(t.unsqueeze(0) if t.ndim == 0 else (t.view(-1) if t.ndim > 1 else t))
    return bags, coords, sizes, torch.stack(tg)


def collate_multitarget(items):
    bags, coords, sizes = _stack_items(items)
    keys: list = []
    for it in items:
        keys += [k for k in it[3] if k not in keys]
    return bags, coords, sizes, {k: torch.stack([it[3][k] for it in items if k in it[3]]) for k in keys}


def tile_bag_dataloader(*, patient_data: Sequence[PatientData], bag_size: int | None, task: str, categories: Sequence[str] | None = None, batch_size: int,
                        shuffle: bool, num_workers: int, transform: Callable[[torch.Tensor], torch.Tensor] | None, worker_init_fn=None, device=None):
    """-> (DataLoader of (bags, coords, bag_sizes, targets), categories).  Sampling is random when the loader shuffles (training) and equidistant
    when it does not (validation / deployment), like the reference (:121)."""
    targets, cats = parse_targets(patient_data=patient_data, task=task, categories=categories)
    multi = isinstance(targets, list)
    ds = BagDataset(bags=[p.feature_files for p in patient_data], bag_size=bag_size, ground_truths=targets, transform=transform, deterministic=not shuffle,
                    device=device)
    dl = DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, collate_fn=collate_multitarget if multi else collate_to_tuple,
                    worker_init_fn=worker_init_fn, persistent_workers=num_workers > 0)
    return dl, cats


# ---- loss weights -----------------------------------------------------------------------------------------------------------------------------
def class_weights(ground_truths, categories: Sequence[str] | Mapping[str, Sequence[str]] | None = None):
    """Inverse class frequency, normalised to sum 1: w_c = (N / n_c) / sum_c' (N / n_c') over the one-hot training targets (train.py:594-600); a
    dict of weight vectors for multi-target ground truths (:578-590).  Raises on a single category, warns about categories with fewer than 16
    samples (:602-620)."""
    def inv_freq(onehot: torch.Tensor) -> torch.Tensor:
        counts = onehot.sum(dim=0)
        w = counts.sum() / counts
        return w / w.sum()
    if isinstance(ground_truths, list):
        return {k: inv_freq(torch.stack([g[k] for g in ground_truths])) for k in ground_truths[0]}
    counts = ground_truths.sum(dim=0)
    if categories is not None:
        if len(categories) <= 1:
            raise ValueError(f"not enough categories to train on: {categories}")
        few = {c: int(n) for c, n in zip(categories, counts.reshape(-1).tolist()) if n < 16}
        if few:
            _log.warning(f"Some categories do not have enough samples to meaningfully train a model: {few}. You may want to consider removing these "
                         "categories; the model will likely overfit on the few samples available.")
    return inv_freq(ground_truths)


__all__ = ["PatientData", "parse_targets", "BagDataset", "fixed_size_bag", "collate_to_tuple", "collate_multitarget", "tile_bag_dataloader", "class_weights",
           "MAX_OPEN_FILES"]
