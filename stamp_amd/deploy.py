"""`stamp deploy` on the HIP heads: the prediction step and the CSV shaping around the MIL forward.

Restates what the reference does AROUND `model(bags, coords=..., mask=...)`:
  * `_predict` (src/stamp/modeling/deploy.py:390-456): eval mode, one forward per patient batch, outputs concatenated in patient order;
    classification -> `softmax(dim=1)` (:447-448), survival -> `squeeze(-1)` (:449-450), regression raw.
  * `_to_prediction_df` single-target (:563-590), `_to_regression_prediction_df` (:593-637), `_to_survival_prediction_df` (:640-691):
    column names, row order and the per-patient `loss` as the reference computes it -- including that the classification loss is
    `cross_entropy` applied to the PROBABILITIES (the softmax output is fed to a function that applies log-softmax again, :577-583).

Parity: pinned.  The reference module cannot be imported here (it needs `lightning` and `h5py`), so `tools/make_golden.py::golden_deploy_tables`
executes its three table functions by name out of the file and commits their output for fixed inputs (`tests/golden/deploy_tables.json`;
`tests/test_cpu_deploy.py` compares cell by cell).  The model is any of `stamp_amd.mil`'s heads (or any module with the reference's forward
signature); with `torch.no_grad()` + `.eval()` they take the forward-only HIP path.
"""
from __future__ import annotations

from collections.abc import Iterable, Mapping, Sequence

import numpy as np
import pandas as pd
import torch
import torch.nn.functional as F

__all__ = ["model_from_checkpoint", "predict_", "to_prediction_df", "to_regression_prediction_df", "to_survival_prediction_df"]


def model_from_checkpoint(ckpt: Mapping) -> torch.nn.Module:
    """The HIP head for a STAMP checkpoint's `{"state_dict", "hyper_parameters"}` (an already unpickled Lightning checkpoint; the reference
    reads it in `load_model_from_ckpt`, deploy.py:49-58).  The backbone sits under `model.` in the Lightning module's state_dict
    (`models/__init__.py:213-215`) and is built as `model_class(dim_input=, dim_output=, **params)` with `params` the hyper-parameters that
    appear in the class's signature (`_get_model_params` / `_build_backbone`, `models/__init__.py:113-131`); `dim_output` is the number of
    categories for classification (`:213`) and 1 for regression / survival.  `model_name` selects the class as the reference's registry
    does (`registry.py:18-25, 43-75`)."""
    import inspect

    from . import mil

    hp = dict(ckpt["hyper_parameters"])
    name = str(getattr(hp.get("model_name"), "value", hp.get("model_name"))).lower()
    if name == "barspoon" or ("category_weights" in hp and "d_model" in hp):
        # LitEncDecTransformer (models/__init__.py:857-899): one head per target, sized by its `category_weights` entry; deploy forward only here
        from .barspoon import EncDecTransformer
        kw = {k: hp[k] for k in ("d_model", "num_encoder_heads", "num_decoder_heads", "num_encoder_layers", "num_decoder_layers", "dim_feedforward", "positional_encoding")
              if k in hp}
        model = EncDecTransformer(int(hp["dim_input"]), {t: len(w) for t, w in hp["category_weights"].items()}, **kw)
        model.load_state_dict({k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")})
        return model.eval()
    classes = {"vit": mil.VisionTransformer, "trans_mil": mil.TransMIL, "mlp": mil.MLP, "linear": mil.Linear}
    if name not in classes:
        raise ValueError(f"model_name {name!r} has no HIP head (available: {sorted(classes) + ['barspoon']})")
    cls = classes[name]
    task = hp.get("task", "classification")
    dim_output = len(hp["categories"]) if task == "classification" else 1
    keys = [k for k in inspect.signature(cls.__init__).parameters if k not in ("self", "dim_input", "dim_output")]
    model = cls(dim_input=int(hp["dim_input"]), dim_output=dim_output, **{k: hp[k] for k in keys if k in hp})
    sd = {k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
    model.load_state_dict(sd)          # strict: a key mismatch means a different architecture
    return model.eval()


@torch.no_grad()
def predict_(model: torch.nn.Module, batches: Iterable, patient_ids: Sequence[str], *, task: str, device="cuda") -> dict[str, torch.Tensor]:
    """batches: iterable of (bags, coords, bag_sizes, targets) as the reference's test DataLoader yields them (full bags, batch 1,
    `modeling/data.py:255-277`); only bags / coords are used (`Lit*.predict_step`, `models/__init__.py:302-313`: `mask=None`).
    Returns patient -> prediction on the CPU: class probabilities (classification), raw value (regression), risk score (survival)."""
    if task not in ("classification", "regression", "survival"):
        raise ValueError(f"unknown task {task!r}")
    model = model.to(device).eval()
    if hasattr(model, "target_labels") and hasattr(model, "class_tokens"):
        # multi-target head (barspoon): `LitMilClassificationMixin.predict_step` returns softmax(logits) per target (barspoon.py:333-344) and the
        # reference's `_predict` applies softmax to what it gets AGAIN when the task is classification (deploy.py:416-438) -- reproduced, not repaired
        per: dict[str, list[torch.Tensor]] = {}
        for bags, coords, *_ in batches:
            out = model(bags.to(device), coords.to(device))
            for t, v in out.items():
                per.setdefault(t, []).append(torch.softmax(v.float(), 1).cpu())
        if not per:
            return {}
        cat = {t: torch.cat(v, dim=0) for t, v in per.items()}
        if task == "classification":
            cat = {t: torch.softmax(v, dim=1) for t, v in cat.items()}
        n = next(iter(cat.values())).shape[0]
        return {pid: {t: cat[t][i] for t in cat} for i, pid in enumerate(list(patient_ids)[:n])}
    outs = []
    for bags, coords, *_ in batches:
        out = model(bags.to(device), coords=None if coords is None else coords.to(device), mask=None)
        outs.append(out.float().cpu())
    if not outs:
        return {}
    raw = torch.cat(outs, dim=0)
    if task == "classification":
        raw = torch.softmax(raw, dim=1)
    elif task == "survival":
        raw = raw.squeeze(-1)
    if raw.shape[0] != len(patient_ids):
        raise ValueError(f"{raw.shape[0]} predictions for {len(patient_ids)} patients")
    return {pid: raw[i] for i, pid in enumerate(patient_ids)}


def to_prediction_df(*, categories: Sequence, patient_to_ground_truth: Mapping, predictions: Mapping[str, torch.Tensor], patient_label: str,
                     ground_truth_label: str) -> pd.DataFrame:
    """patient | ground truth | pred | <ground_truth_label>_<category> ... | loss, sorted by loss (deploy.py:563-590).  Multi-target predictions
    (patient -> {target: probabilities}, barspoon): patient | <target> ... | pred_<t> | <t>_<category> ... | loss = sum over the targets with a
    known ground truth, in prediction order, unsorted (:479-558); `categories`: {target: [category, ...]}; a missing list is inferred from the
    ground truths (sorted)."""
    first = next(iter(predictions.values())) if len(predictions) else None
    if isinstance(first, dict):
        targets = list(first.keys())
        cats_map = dict(categories) if isinstance(categories, dict) else {}
        if not isinstance(categories, dict) and isinstance(categories, Sequence):
            try:
                cats_map = {t: list(categories[i]) for i, t in enumerate(targets)}
            except Exception:  # noqa: BLE001
                cats_map = {}
        if any(t not in cats_map for t in targets):
            inferred = {t: set() for t in targets}
            for gt in patient_to_ground_truth.values():
                if isinstance(gt, dict):
                    for t in targets:
                        if gt.get(t) is not None:
                            inferred[t].add(gt.get(t))
            for t in targets:
                cats_map.setdefault(t, sorted(inferred[t]))
        rows = []
        for pid, pd_ in predictions.items():
            gt_entry = patient_to_ground_truth.get(pid)
            row: dict = {patient_label: pid}
            for t in targets:
                row[t] = gt_entry.get(t) if isinstance(gt_entry, dict) else gt_entry
            total, has = 0.0, False
            for t in targets:
                probs = pd_[t].detach().cpu()
                cats_t = cats_map.get(t, [])
                if probs.numel() == 1:
                    row[f"pred_{t}"] = float(probs.item())
                else:
                    idx = int(probs.argmax().item())
                    row[f"pred_{t}"] = cats_t[idx] if idx < len(cats_t) else idx
                for i_cat, c in enumerate(cats_t):
                    row[f"{t}_{c}"] = float(probs[i_cat].item()) if i_cat < probs.shape[0] else None
                if isinstance(gt_entry, dict) and gt_entry.get(t) is not None:
                    try:
                        ti = int(np.where(np.array(cats_t) == gt_entry.get(t))[0][0])
                        total += F.cross_entropy(probs.reshape(1, -1), torch.tensor([ti])).item()
                        has = True
                    except Exception:  # noqa: BLE001
                        pass
            row["loss"] = total if has else None
            rows.append(row)
        return pd.DataFrame(rows)
    cats = list(categories)
    rows = []
    for pid, prediction in predictions.items():
        gt = patient_to_ground_truth.get(pid)
        row = {patient_label: pid, ground_truth_label: gt, "pred": cats[int(prediction.argmax())]}
        for i_cat, category in enumerate(cats):
            row[f"{ground_truth_label}_{category}"] = float(prediction[i_cat].item())
        row["loss"] = (F.cross_entropy(prediction.reshape(1, -1), torch.tensor(np.where(np.array(cats) == gt)[0])).item()
                       if gt is not None else None)
        rows.append(row)
    return pd.DataFrame(rows).sort_values(by="loss")


def to_regression_prediction_df(*, patient_to_ground_truth: Mapping, predictions: Mapping[str, torch.Tensor], patient_label: str,
                                ground_truth_label: str) -> pd.DataFrame:
    """patient | ground truth | pred | loss (per-sample L1 when the ground truth is a number), sorted by loss, missing last (:593-637)."""
    rows = []
    for pid, prediction in predictions.items():
        gt = patient_to_ground_truth.get(pid)
        has = gt is not None and str(gt).lower() != "nan" and prediction.numel() == 1
        rows.append({patient_label: pid, ground_truth_label: gt,
                     "pred": float(prediction.flatten().item()) if prediction.numel() == 1 else prediction.cpu().tolist(),
                     "loss": F.l1_loss(prediction.flatten(), torch.tensor([float(gt)], dtype=prediction.dtype)).item() if has else None})
    return pd.DataFrame(rows).sort_values(by="loss", na_position="last")


def to_survival_prediction_df(*, patient_to_ground_truth: Mapping, predictions: Mapping[str, torch.Tensor], patient_label: str,
                              time_label: str = "time", status_label: str = "event", cut_off: float | None = None) -> pd.DataFrame:
    """patient | pred_score | time | event (| cut_off=<c>), in prediction order; ground truth is a (time, event) pair or unknown (:640-691)."""
    rows = []
    for pid, pred in predictions.items():
        pred = pred.detach().flatten()
        gt = patient_to_ground_truth.get(pid)
        row = {patient_label: pid, "pred_score": float(pred.item()) if pred.numel() == 1 else pred.cpu().tolist()}
        row[time_label], row[status_label] = gt if isinstance(gt, (tuple, list)) and len(gt) == 2 else (None, None)
        rows.append(row)
    df = pd.DataFrame(rows)
    if cut_off is not None:
        df[f"cut_off={cut_off}"] = None
    return df
