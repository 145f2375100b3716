"""The shaping around the MIL forward that `stamp deploy` does (stamp_amd/deploy.py; reference src/stamp/modeling/deploy.py:390-691),
checked against the reference's formulas evaluated directly (no fixtures: the reference module needs lightning + h5py to import)."""
import math

import pandas as pd
import torch

from stamp_amd.deploy import predict_, to_prediction_df, to_regression_prediction_df, to_survival_prediction_df


class _Head(torch.nn.Module):
    """Stand-in with the reference heads' forward signature: mean over tiles, then a Linear."""

    def __init__(self, f, c):
        super().__init__()
        self.lin = torch.nn.Linear(f, c)

    def forward(self, bags, *, coords=None, mask=None):
        assert mask is None
        return self.lin(bags.float().mean(1))


def test_predict_tasks_and_order():
    torch.manual_seed(0)
    head = _Head(8, 3)
    bags = [torch.randn(1, n, 8) for n in (5, 9, 2)]
    batches = [(b, torch.zeros(1, b.shape[1], 2), None, None) for b in bags]
    pids = ["p2", "p0", "p1"]
    out = predict_(head, batches, pids, task="classification", device="cpu")
    assert list(out) == pids
    for b, pid in zip(bags, pids):
        want = torch.softmax(head(b), dim=1)[0]
        assert torch.allclose(out[pid], want.detach(), atol=1e-6) and abs(out[pid].sum().item() - 1) < 1e-6
    surv = predict_(_Head(8, 1), batches, pids, task="survival", device="cpu")
    assert all(v.dim() == 0 for v in surv.values())
    reg = predict_(_Head(8, 1), batches, pids, task="regression", device="cpu")
    assert all(v.shape == (1,) for v in reg.values())


def test_classification_table_columns_loss_and_order():
    cats = ["mut", "wt"]
    preds = {"a": torch.tensor([0.9, 0.1]), "b": torch.tensor([0.3, 0.7]), "c": torch.tensor([0.6, 0.4])}
    gts = {"a": "wt", "b": "wt", "c": None}
    df = to_prediction_df(categories=cats, patient_to_ground_truth=gts, predictions=preds, patient_label="PATIENT", ground_truth_label="KRAS")
    assert list(df.columns) == ["PATIENT", "KRAS", "pred", "KRAS_mut", "KRAS_wt", "loss"]
    # the reference feeds PROBABILITIES to cross_entropy: loss = -log softmax(p)[target]
    la = -math.log(math.exp(0.1) / (math.exp(0.9) + math.exp(0.1)))
    lb = -math.log(math.exp(0.7) / (math.exp(0.3) + math.exp(0.7)))
    assert list(df["PATIENT"]) == ["b", "a", "c"]                   # sorted by loss, missing last
    assert abs(df.iloc[0]["loss"] - lb) < 1e-6 and abs(df.iloc[1]["loss"] - la) < 1e-6 and pd.isna(df.iloc[2]["loss"])
    assert list(df["pred"]) == ["wt", "mut", "mut"]
    assert abs(df.iloc[1]["KRAS_mut"] - 0.9) < 1e-6


def test_regression_and_survival_tables():
    preds = {"a": torch.tensor([2.5]), "b": torch.tensor([1.0]), "c": torch.tensor([0.5])}
    df = to_regression_prediction_df(patient_to_ground_truth={"a": 2.0, "b": None, "c": "nan"}, predictions=preds, patient_label="P",
                                     ground_truth_label="age")
    assert list(df.columns) == ["P", "age", "pred", "loss"]
    assert list(df["P"])[0] == "a" and abs(df.iloc[0]["loss"] - 0.5) < 1e-6 and df["loss"].isna().sum() == 2
    sdf = to_survival_prediction_df(patient_to_ground_truth={"a": (302.0, 1), "b": "302 dead"}, predictions={"a": torch.tensor(0.3), "b": torch.tensor([1.5])},
                                    patient_label="P", cut_off=0.7)
    assert list(sdf.columns) == ["P", "pred_score", "time", "event", "cut_off=0.7"]
    assert sdf.iloc[0]["time"] == 302.0 and sdf.iloc[0]["event"] == 1 and pd.isna(sdf.iloc[1]["time"])    # pandas stores the unknown as NaN
    assert abs(sdf.iloc[1]["pred_score"] - 1.5) < 1e-6


def test_model_from_checkpoint_builds_the_head_and_loads_the_backbone():
    """A Lightning-style checkpoint dict (backbone under `model.`, constructor arguments among the hyper-parameters next to unrelated metadata)
    -> the HIP head with exactly those weights, in eval mode; unknown model names and foreign keys fail loudly."""
    import pytest

    from stamp_amd.deploy import model_from_checkpoint
    from stamp_amd.mil import VisionTransformer

    torch.manual_seed(1)
    src = VisionTransformer(dim_output=3, dim_input=96, dim_model=64, n_layers=1, n_heads=2, dim_feedforward=64, dropout=0.1, use_alibi=True)
    ckpt = {"state_dict": {**{f"model.{k}": v.clone() for k, v in src.state_dict().items()}, "valid_auroc.something": torch.zeros(1)},
            "hyper_parameters": {"model_name": "vit", "task": "classification", "categories": ["a", "b", "c"], "dim_input": 96, "dim_model": 64, "n_layers": 1,
                                 "n_heads": 2, "dim_feedforward": 64, "dropout": 0.1, "use_alibi": True, "total_steps": 10, "max_lr": 1e-4, "ground_truth_label": "x"}}
    m = model_from_checkpoint(ckpt)
    assert isinstance(m, VisionTransformer) and not m.training
    for k, v in src.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    with pytest.raises(ValueError, match="no HIP head"):
        model_from_checkpoint({**ckpt, "hyper_parameters": {**ckpt["hyper_parameters"], "model_name": "nonesuch"}})
    # barspoon checkpoints (LitEncDecTransformer, models/__init__.py:857-899): heads sized by `category_weights`
    from stamp_amd.barspoon import EncDecTransformer
    bsrc = EncDecTransformer(40, {"KRAS": 2, "MSI status": 3}, d_model=64, num_encoder_heads=1, num_decoder_heads=1, num_encoder_layers=1, num_decoder_layers=1,
                             dim_feedforward=128, positional_encoding=False)
    bck = {"state_dict": {f"model.{k}": v.clone() for k, v in bsrc.state_dict().items()},
           "hyper_parameters": {"task": "classification", "dim_input": 40, "category_weights": {"KRAS": torch.ones(2), "MSI status": torch.ones(3)}, "d_model": 64,
                                "num_encoder_heads": 1, "num_decoder_heads": 1, "num_encoder_layers": 1, "num_decoder_layers": 1, "dim_feedforward": 128,
                                "positional_encoding": False, "learning_rate": 1e-4, "categories": {"KRAS": ["wt", "mut"], "MSI status": ["a", "b", "c"]}}}
    bm = model_from_checkpoint(bck)
    assert isinstance(bm, EncDecTransformer) and not bm.training and list(bm.target_labels) == ["KRAS", "MSI status"] and not bm.positional_encoding
    for k, v in bsrc.state_dict().items():
        assert torch.equal(bm.state_dict()[k], v), k
    bad = {**ckpt, "state_dict": {**ckpt["state_dict"], "model.not_a_key": torch.zeros(1)}}
    with pytest.raises(RuntimeError):
        model_from_checkpoint(bad)


def test_tables_equal_the_references_own_functions():
    """The three prediction tables against CSV text produced by the reference's own `_to_prediction_df` (single- and multi-target, with given and with
    inferred category lists), `_to_regression_prediction_df` and `_to_survival_prediction_df` on the same inputs (tests/golden/deploy_tables.json,
    tools/make_golden.py::golden_deploy_tables): same columns, same row order, same values."""
    import io
    import json
    from pathlib import Path

    z = json.loads((Path(__file__).parent / "golden" / "deploy_tables.json").read_text())

    def same(df, csv):
        ref = pd.read_csv(io.StringIO(csv))
        got = pd.read_csv(io.StringIO(df.to_csv(index=False)))
        assert list(got.columns) == list(ref.columns), (list(got.columns), list(ref.columns))
        pd.testing.assert_frame_equal(got, ref, check_exact=False, rtol=1e-6, atol=1e-7)

    s = z["single"]
    same(to_prediction_df(categories=s["categories"], patient_to_ground_truth=s["gts"], predictions={k: torch.tensor(v) for k, v in s["preds"].items()},
                          patient_label="PATIENT", ground_truth_label="KRAS"), s["csv"])
    m = z["multi"]
    mp = {k: {t: torch.tensor(v) for t, v in d.items()} for k, d in m["preds"].items()}
    same(to_prediction_df(categories=m["categories"], patient_to_ground_truth=m["gts"], predictions=mp, patient_label="PATIENT", ground_truth_label=["KRAS", "MSI status"]),
         m["csv"])
    same(to_prediction_df(categories=[], patient_to_ground_truth=m["gts"], predictions=mp, patient_label="PATIENT", ground_truth_label=None), z["multi_inferred_csv"])
    r = z["regression"]
    same(to_regression_prediction_df(patient_to_ground_truth=r["gts"], predictions={k: torch.tensor(v) for k, v in r["preds"].items()}, patient_label="P",
                                     ground_truth_label="age"), r["csv"])
    sv = z["survival"]
    same(to_survival_prediction_df(patient_to_ground_truth={k: (tuple(v) if isinstance(v, list) else v) for k, v in sv["gts"].items()},
                                   predictions={k: torch.tensor(v) for k, v in sv["preds"].items()}, patient_label="P", cut_off=0.7), sv["csv"])


def test_predict_multi_target_head_reproduces_the_references_double_softmax():
    class _Multi(torch.nn.Module):                       # the barspoon head's surface: forward(tokens, positions) -> {target: logits}
        target_labels = ["A", "B"]
        class_tokens = None

        def forward(self, x, pos):
            m = x.float().mean(1)
            return {"A": m[:, :2], "B": m[:, 2:5]}

    bags = [torch.randn(1, n, 8) for n in (4, 7)]
    out = predict_(_Multi(), [(b, torch.zeros(1, b.shape[1], 2), None, None) for b in bags], ["x", "y"], task="classification", device="cpu")
    assert list(out) == ["x", "y"] and set(out["x"]) == {"A", "B"}
    want = torch.softmax(torch.softmax(bags[1].mean(1)[:, 2:5], 1), 1)[0]          # predict_step's softmax, then _predict's (deploy.py:416-438)
    assert torch.allclose(out["y"]["B"], want, atol=1e-6)
