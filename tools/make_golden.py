"""Generate golden vectors under tests/golden/ by IMPORTING the reference's own model code.

Runs only in the build container (needs /root/reference).  Nothing of the reference travels: the fixtures are
data -- seeded inputs, the state_dict the reference module was initialised with, and the outputs the
reference module produced -- stored as .npz.  Re-run:  python tools/make_golden.py

Shims (arithmetic-free): typing.assert_never / enum.StrEnum / hashlib.file_digest
polyfills for Python 3.10, stub modules for jaxtyping / beartype / gdown, a synthetic `stamp` namespace so
sub-modules resolve without running the package __init__s (which import h5py / openslide).
"""
from __future__ import annotations

import ast
import enum
import hashlib
import sys
import types
import typing
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/src/stamp")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def install_shims() -> None:
    import typing_extensions

    if not hasattr(typing, "assert_never"):
        typing.assert_never = typing_extensions.assert_never
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)
        enum.StrEnum = StrEnum
    if not hasattr(hashlib, "file_digest"):
        def file_digest(f, algo):
            h = hashlib.new(algo) if isinstance(algo, str) else algo()
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
            return h
        hashlib.file_digest = file_digest

    class _Sub:
        def __getitem__(self, item):
            return typing.Any

    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Bool", "Integer", "Int", "Shaped"):
        setattr(jt, n, _Sub())
    jt.jaxtyped = lambda *a, **k: (lambda f: f)
    sys.modules["jaxtyping"] = jt
    bt = types.ModuleType("beartype")
    bt.beartype = lambda f=None, **k: f if f is not None else (lambda g: g)
    sys.modules["beartype"] = bt
    sys.modules["gdown"] = types.ModuleType("gdown")
    if not hasattr(torch.nn, "Buffer"):
        raise RuntimeError("torch.nn.Buffer missing")
    # synthetic namespace packages
    for name, path in (("stamp", REF), ("stamp.modeling", REF / "modeling"),
                       ("stamp.modeling.models", REF / "modeling" / "models")):
        m = types.ModuleType(name)
        m.__path__ = [str(path)]
        sys.modules[name] = m


def load_by_path(modname: str, path: Path):
    import importlib.util

    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def exec_defs(path: Path, names: set[str], glb: dict, inherit_future: bool = True) -> dict:
    """exec only the named top-level class/function definitions of a reference file (its module top imports
    things that are absent here).  inherit_future=False: compile WITHOUT this file's `from __future__ import annotations` (a dataclass that
    uses `_: KW_ONLY` needs the annotation evaluated)."""
    tree = ast.parse(path.read_text())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    missing = names - {n.name for n in body}
    assert not missing, missing
    code = compile(ast.Module(body=body, type_ignores=[]), str(path), "exec", dont_inherit=not inherit_future)
    exec(code, glb)
    return glb


def exec_method(path: Path, method: str, glb: dict):
    """exec ONE method of a class of a reference file as a plain function (decorators dropped): for static methods of classes whose
    module cannot be imported here (lightning)."""
    tree = ast.parse(path.read_text())
    for cls in tree.body:
        if isinstance(cls, ast.ClassDef):
            for fn in cls.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == method:
                    fn.decorator_list = []
                    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(path), "exec"), glb)
                    return glb[method]
    raise AssertionError(f"{method} not found in {path}")


def sd_np(module: torch.nn.Module, prefix: str = "w:") -> dict[str, np.ndarray]:
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}      # copy: buffers are updated in place later


def save(name: str, **arrs) -> None:
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / name, **arrs)
    size = (OUT / name).stat().st_size
    print(f"wrote {name}: {size/1024:.1f} KiB, keys={len(arrs)}")


def golden_chief() -> None:
    import torch.nn as nn
    import torch.nn.functional as F

    glb = {"nn": nn, "torch": torch, "F": F}
    exec_defs(REF / "encoding" / "encoder" / "chief.py",
              {"CHIEFModel", "Attn_Net_Gated", "Attn_Net", "Att_Head", "initialize_weights"}, glb)
    for tag, size_arg, N in (("small", "small", 300), ("xs", "xs", 77)):
        torch.manual_seed(100 + N)
        model = glb["CHIEFModel"](size_arg=size_arg, dropout=True, n_classes=2).eval()
        # the reference's initialize_weights gives xavier weights and zero biases; perturb biases so they matter
        with torch.no_grad():
            for p in model.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        Fdim = model.size_dict[size_arg][0]
        x = torch.randn(N, Fdim) * 0.7
        with torch.no_grad():
            out = model(x)
        save(f"chief_gated_attention_{tag}.npz", x=x.numpy(), wsi_feature=out["WSI_feature"].numpy(),
             attention_raw=out["attention_raw"].numpy(), size_arg=np.array(size_arg),
             **{k: v for k, v in sd_np(model).items() if k.startswith("w:attention_net.")})  # only the used path


def golden_eagle() -> None:
    """EAGLE slide encoder: the reference's own `_generate_slide_embedding` (run as a plain function on a stand-in `self` holding the reference's
    CHIEFModel) and its coordinate alignment helper, on seeded inputs."""
    import torch.nn as nn
    import torch.nn.functional as F

    glb = {"nn": nn, "torch": torch, "F": F, "np": np}
    exec_defs(REF / "encoding" / "encoder" / "chief.py", {"CHIEFModel", "Attn_Net_Gated", "Attn_Net", "Att_Head", "initialize_weights"}, glb)
    from collections import defaultdict, deque
    eg = {"torch": torch, "np": np, "Tensor": torch.Tensor, "DeviceLikeType": object, "defaultdict": defaultdict, "deque": deque}
    gen = exec_method(REF / "encoding" / "encoder" / "eagle.py", "_generate_slide_embedding", eg)
    exec_defs(REF / "encoding" / "encoder" / "eagle.py", {"_align_vir2_to_ctp_by_coords"}, eg)
    align = eg["_align_vir2_to_ctp_by_coords"]
    out = {}
    torch.manual_seed(500)
    model = glb["CHIEFModel"](size_arg="xs", dropout=True, n_classes=2).eval()
    Fdim = model.size_dict["xs"][0]
    with torch.no_grad():           # values exactly representable in 16 bits: the fixture compresses to a fraction of its size
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            p.copy_(p.bfloat16().float())
    out.update({k: v for k, v in sd_np(model).items() if k.startswith("w:attention_net.")})
    for tag, N in (("a", 200), ("b", 9)):                       # b: fewer tiles than the 25 EAGLE keeps
        x = (torch.randn(N, Fdim) * 0.7).half().float()
        agg = torch.randn(N, 64).half().float()
        me = types.SimpleNamespace(model=model)
        emb = gen(me, x, "cpu", agg)
        with torch.no_grad():
            araw = model(x)["attention_raw"].squeeze(0)
        out.update({f"{tag}_x": x.numpy(), f"{tag}_agg": agg.numpy(), f"{tag}_emb": emb, f"{tag}_top": torch.topk(araw, min(25, N))[1].numpy(),
                    f"{tag}_araw": araw.numpy()})
    # alignment: a shuffled copy of a coordinate set that contains one duplicate pair
    rng = np.random.default_rng(7)
    ref = (rng.integers(0, 40, (60, 2)) * 256.0).astype(np.float32)
    ref[17] = ref[3]
    perm0 = rng.permutation(60)
    other = ref[perm0] + rng.uniform(-2e-6, 2e-6, (60, 2)).astype(np.float32)
    feats = torch.arange(60, dtype=torch.float32).unsqueeze(1).repeat(1, 3)[perm0]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        al_feats, al_coords = align(ref, other, feats, 5)
    out.update(al_ref=ref, al_other=other, al_ids=feats[:, 0].numpy().astype(np.int64), al_rows=al_feats[:, 0].numpy().astype(np.int64), al_coords=al_coords)
    save("eagle.npz", **out)


def golden_barspoon() -> None:
    """barspoon head: the reference's own `EncDecTransformer` (its module imports lightning / torchmetrics: only the class and `sanitize` are
    executed), eval mode, two geometries: the defaults' shape in small (2 + 2 layers) and a single-target single-layer one without positional
    encoding."""
    import re

    import torch.nn as nn
    import torch.nn.functional as F

    glb = {"nn": nn, "torch": torch, "F": F, "re": re}
    exec_defs(REF / "modeling" / "models" / "barspoon.py", {"EncDecTransformer", "sanitize"}, glb)
    out = {}
    for tag, kw, targets, Fd, B, T in (("a", dict(d_model=128, num_encoder_heads=2, num_decoder_heads=2, num_encoder_layers=2, num_decoder_layers=2, dim_feedforward=256),
                                        {"KRAS": 2, "MSI status": 3, "grade-x": 4}, 96, 2, 150),
                                       ("b", dict(d_model=64, num_encoder_heads=1, num_decoder_heads=1, num_encoder_layers=1, num_decoder_layers=1, dim_feedforward=128,
                                                  positional_encoding=False), {"t": 2}, 40, 1, 33)):
        torch.manual_seed(900 + T)
        model = glb["EncDecTransformer"](Fd, targets, **kw).eval()
        with torch.no_grad():           # values exactly representable in 16 bits: the fixture compresses
            for p in model.parameters():
                if p.dim() == 1:
                    p.add_(0.05 * torch.randn_like(p))
                p.copy_(p.bfloat16().float())
        x = torch.randn(B, T, Fd).half().float()
        pos = (torch.rand(B, T, 2) * 30000.0).half().float()
        with torch.no_grad():
            logits = model(x, pos)
        out.update({f"{tag}_x": x.numpy(), f"{tag}_pos": pos.numpy(), f"{tag}_targets": np.array(list(targets)), f"{tag}_nout": np.array(list(targets.values()))})
        out.update({f"{tag}_logits_{j}": logits[t].numpy() for j, t in enumerate(targets)})
        out.update({f"{tag}_{k}": v for k, v in sd_np(model).items()})
        out[f"{tag}_hparams"] = np.array([kw["d_model"], kw["num_encoder_heads"], kw["num_decoder_heads"], kw["num_encoder_layers"], kw["num_decoder_layers"],
                                          kw["dim_feedforward"], int(kw.get("positional_encoding", True))])
    save("barspoon.npz", **out)


def golden_ticon() -> None:
    """TICON tile contextualiser: the reference's own `EncoderDecoder` (ticon.py; the module top imports timm / huggingface_hub: only the model
    classes are executed) at a small width, called exactly as `HOptimusTICON.forward` calls it -- one token per tile, zero coordinates."""
    import math
    from collections.abc import Callable, Mapping
    from functools import partial
    from typing import Any

    import torch.nn as nn

    glb = {"nn": nn, "torch": torch, "math": math, "Tensor": torch.Tensor, "Float": sys.modules["jaxtyping"].Float, "Callable": Callable, "Mapping": Mapping,
           "Any": Any, "partial": partial}
    exec_defs(REF / "preprocessing" / "extractor" / "ticon.py",
              {"LayerScale", "Mlp", "ProjectionMlp", "get_slopes", "scaled_dot_product_attention_custom", "Attention", "NaiveResidual", "EfficientResidual", "Block",
               "Transformer", "EncoderDecoder", "_init_weights"}, glb)
    torch.manual_seed(77)
    cfg = dict(transformers_kwargs={"embed_dim": 96, "drop_path_rate": 0.0, "block_kwargs": {"attn_kwargs": {"num_heads": 6}}}, encoder_kwargs={"depth": 3},
               decoder_kwargs={"depth": 1}, in_dims=[48, 128], tile_encoder_keys=["conchv15", "hoptimus1"], num_decoders=1, decoder_out_dims=[48, 128])
    model = glb["EncoderDecoder"](**cfg).init_weights().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():                       # the reference initialises biases to 0 and LayerScale to 1: make every parameter matter
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            p.copy_(p.bfloat16().float())
    out = {}
    for key, d_in in (("hoptimus1", 128), ("conchv15", 48)):
        emb = torch.randn(9, d_in).half().float()
        with torch.no_grad():
            y = model(x=emb.unsqueeze(1), relative_coords=torch.zeros(9, 1, 2), tile_encoder_key=key).squeeze(1)
        out[f"emb_{key}"], out[f"out_{key}"] = emb.numpy(), y.numpy()
    out.update({k: v for k, v in sd_np(model).items() if ".decoder_" not in k and "output_proj" not in k and "mask_dict" not in k})
    save("ticon.npz", **out)


def golden_keep_head() -> None:
    """KEEP's image head: the reference's own `KEEPImageModel.encode_image` (keep.py:25-50) with the timm trunk replaced by the identity (timm is not
    in this image; the trunk is pinned elsewhere) -- pins visual_head + the L2 normalisation."""
    import torch.nn as nn

    class _Trunk(nn.Module):
        num_features = 128

        def forward(self, x):
            return x

    timm_stub = types.SimpleNamespace(create_model=lambda *a, **k: _Trunk())
    glb = {"nn": nn, "torch": torch, "timm": timm_stub}
    exec_defs(REF / "preprocessing" / "extractor" / "keep.py", {"KEEPImageModel"}, glb)
    torch.manual_seed(31)
    model = glb["KEEPImageModel"](vision_config={"img_size": 224, "patch_size": 16, "init_values": 1e-5, "num_classes": 0}, projection_dim=96).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.bfloat16().float())
    feats = torch.randn(11, 128).half().float()
    feats[3] = 0.0                                                  # an all-zero feature row: normalize's eps branch after the biases
    with torch.no_grad():
        out = model(feats)
    save("keep_head.npz", feats=feats.numpy(), out=out.numpy(), **{k: v for k, v in sd_np(model).items() if k.startswith("w:visual_head.")})


def golden_plip() -> None:
    """PLIP's vision tower: the installed `transformers` CLIPModel itself (the package the reference's plip.py calls), randomly initialised at a small
    width, `get_image_features` on transformed tiles."""
    import logging

    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from oracle.clip_vision import tiles_to_pixels
    logging.getLogger("transformers").setLevel(logging.ERROR)
    vc = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=224, patch_size=32, projection_dim=64)
    tc = CLIPTextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=100, max_position_embeddings=16, projection_dim=64,
                        bos_token_id=1, eos_token_id=2, pad_token_id=0)
    torch.manual_seed(41)
    model = CLIPModel(CLIPConfig(text_config=tc.to_dict(), vision_config=vc.to_dict(), projection_dim=64)).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("vision_model") or n.startswith("visual_projection"):
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
                p.copy_(p.bfloat16().float())
    tiles = torch.randint(0, 256, (3, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(42))
    with torch.no_grad():
        y = model.get_image_features(tiles_to_pixels(tiles))
    y = y if isinstance(y, torch.Tensor) else y.pooler_output           # transformers 5: a BaseModelOutputWithPooling whose pooler_output is the projected embedding
    import transformers
    save("plip.npz", tiles=tiles.numpy(), image_features=y.numpy(), heads=np.array(2), transformers_version=np.array(transformers.__version__),
         **{k: v for k, v in sd_np(model).items() if k.startswith("w:vision_model") or k.startswith("w:visual_projection")})


def golden_dinov2_hf() -> None:
    """The ViT trunk against an INDEPENDENT third-party implementation that IS in this image: `transformers`' Dinov2Model / Dinov2WithRegistersModel
    (timm, which the reference calls, is not installed; DINOv2 is the architecture of the reference's RedDino / DinoBloom extractors and -- with
    SwiGLU and register tokens -- of its ViT-g family).  Random weights at a small width, three structural variants."""
    import logging

    from transformers import Dinov2Config, Dinov2Model, Dinov2WithRegistersConfig, Dinov2WithRegistersModel

    logging.getLogger("transformers").setLevel(logging.ERROR)
    out = {}
    tiles = torch.randint(0, 256, (3, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(52))
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    px = (tiles.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
    out["tiles"] = tiles.numpy()
    for tag, cls, kw in (("gelu", Dinov2Model, dict(mlp_ratio=2, use_swiglu_ffn=False)), ("swiglu", Dinov2Model, dict(mlp_ratio=4, use_swiglu_ffn=True)),
                         ("reg4", Dinov2WithRegistersModel, dict(mlp_ratio=4, use_swiglu_ffn=True, num_register_tokens=4))):
        ccls = Dinov2WithRegistersConfig if cls is Dinov2WithRegistersModel else Dinov2Config
        cfg = ccls(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, patch_size=14, image_size=224, layerscale_value=0.3, layer_norm_eps=1e-6, **kw)
        torch.manual_seed(60 + len(tag))
        model = cls(cfg).eval()
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() == 1 or "token" in n or "position" in n:
                    p.add_(0.1 * torch.randn_like(p))
                p.copy_(p.bfloat16().float())
            toks = model(px).last_hidden_state
        out[f"{tag}_tokens"] = toks[:, list(range(10)) + [-2, -1]].numpy()          # class token, registers, first and last patches (the fixture stays small)
        out[f"{tag}_token_norms"] = toks.norm(dim=-1).numpy()                            # ... and every token's norm
        out.update({f"{tag}_{k}": v for k, v in sd_np(model).items() if "mask_token" not in k})
    import transformers
    out["transformers_version"] = np.array(transformers.__version__)
    save("dinov2_hf.npz", **out)


def golden_deploy_tables() -> None:
    """The three prediction tables of `stamp deploy`: the reference's own `_to_prediction_df` (single- and multi-target), `_to_regression_prediction_df`,
    `_to_survival_prediction_df` (src/stamp/modeling/deploy.py:459-691; the module imports lightning: only these functions are executed) on fixed
    inputs; stored as the inputs + the CSV text of each resulting table."""
    import json
    from collections.abc import Mapping, Sequence
    from typing import Any, cast

    import pandas as pd
    import torch.nn.functional as F

    glb = {"pd": pd, "np": np, "torch": torch, "F": F, "cast": cast, "Mapping": Mapping, "Sequence": Sequence, "PatientId": str, "GroundTruth": Any, "PandasLabel": str,
           "Category": Any, "SurvivalGroundTruth": Any}
    exec_defs(REF / "modeling" / "deploy.py", {"_to_prediction_df", "_to_regression_prediction_df", "_to_survival_prediction_df"}, glb)
    g = torch.Generator().manual_seed(5)
    out: dict = {}
    cats = ["mut", "wt", "other"]
    preds = {f"p{i}": torch.softmax(torch.randn(3, generator=g), 0) for i in range(6)}
    gts = {"p0": "wt", "p1": "mut", "p2": None, "p3": "other", "p4": "wt", "p5": "mut"}
    df = glb["_to_prediction_df"](categories=cats, patient_to_ground_truth=gts, predictions=preds, patient_label="PATIENT", ground_truth_label="KRAS")
    out["single"] = {"categories": cats, "gts": gts, "preds": {k: v.tolist() for k, v in preds.items()}, "csv": df.to_csv(index=False)}
    mcats = {"KRAS": ["mut", "wt"], "MSI status": ["MSI", "MSS", "unknown"]}
    mpreds = {f"q{i}": {"KRAS": torch.softmax(torch.randn(2, generator=g), 0), "MSI status": torch.softmax(torch.randn(3, generator=g), 0)} for i in range(5)}
    mgts = {"q0": {"KRAS": "wt", "MSI status": "MSS"}, "q1": {"KRAS": None, "MSI status": "MSI"}, "q2": {"KRAS": None, "MSI status": None}, "q3": {"KRAS": "mut", "MSI status": "unknown"}}
    dfm = glb["_to_prediction_df"](categories=mcats, patient_to_ground_truth=mgts, predictions=mpreds, patient_label="PATIENT", ground_truth_label=["KRAS", "MSI status"])
    out["multi"] = {"categories": mcats, "gts": mgts, "preds": {k: {t: v.tolist() for t, v in d.items()} for k, d in mpreds.items()}, "csv": dfm.to_csv(index=False)}
    dfi = glb["_to_prediction_df"](categories=[], patient_to_ground_truth=mgts, predictions=mpreds, patient_label="PATIENT", ground_truth_label=None)     # categories inferred
    out["multi_inferred_csv"] = dfi.to_csv(index=False)
    rpreds = {"a": torch.tensor([2.5]), "b": torch.tensor([1.0]), "c": torch.tensor([0.5]), "d": torch.tensor([-1.25])}
    rgts = {"a": 2.0, "b": None, "c": "nan", "d": 0.75}
    out["regression"] = {"gts": rgts, "preds": {k: v.tolist() for k, v in rpreds.items()},
                         "csv": glb["_to_regression_prediction_df"](patient_to_ground_truth=rgts, predictions=rpreds, patient_label="P", ground_truth_label="age").to_csv(index=False)}
    spreds = {"a": torch.tensor(0.3), "b": torch.tensor([1.5]), "c": torch.tensor([-0.2])}
    sgts = {"a": [302.0, 1], "b": "302 dead", "c": [55.5, 0]}
    out["survival"] = {"gts": sgts, "preds": {k: v.flatten().tolist() for k, v in spreds.items()},
                       "csv": glb["_to_survival_prediction_df"](patient_to_ground_truth={k: (tuple(v) if isinstance(v, list) else v) for k, v in sgts.items()}, predictions=spreds,
                                                                patient_label="P", cut_off=0.7).to_csv(index=False)}
    OUT.mkdir(parents=True, exist_ok=True)
    (OUT / "deploy_tables.json").write_text(json.dumps(out, indent=1))
    print("wrote deploy_tables.json")


def golden_get_coords() -> None:
    """The reference's own feature-file coordinate loader `get_coords` / `get_stride` / `CoordsInfo` (src/stamp/modeling/data.py:726-808, 1150-1161; the module
    imports h5py: a stand-in with just `Dataset` / `File`-like objects is handed to the functions), on the formats it distinguishes: STAMP v2, the current
    one, the historic 224-stride one (by attribute and by stride), the coords-less bypass, a file from a newer STAMP, an un-inferable file."""
    import json
    import logging
    from dataclasses import dataclass
    from typing import cast

    from packaging.version import Version

    class Dataset:                                                   # h5py.Dataset stand-in
        def __init__(self, arr):
            self.arr = np.asarray(arr)
            self.shape = self.arr.shape

        def __getitem__(self, k):
            return self.arr[k]

    class FileStub(dict):
        filename = "stub.h5"

        def __init__(self, datasets, attrs):
            super().__init__({k: Dataset(v) for k, v in datasets.items()})
            self.attrs = dict(attrs)

    h5 = types.SimpleNamespace(Dataset=Dataset, File=FileStub)
    glb = {"np": np, "torch": torch, "h5py": h5, "dataclass": dataclass, "Version": Version, "stamp": types.SimpleNamespace(__version__="2.5.0"), "cast": cast,
           "Tensor": torch.Tensor, "Microns": float, "TilePixels": int, "SlideMPP": float, "_logger": logging.getLogger("golden")}
    tree = ast.parse((REF / "modeling" / "data.py").read_text())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in {"CoordsInfo", "get_coords", "get_stride"}]
    assert len(body) == 3
    for n in body:                                                   # keep @dataclass on CoordsInfo, nothing else is decorated
        exec(compile(ast.Module(body=[n], type_ignores=[]), "data.py", "exec"), glb)
    rng = np.random.default_rng(3)
    grid = np.stack([rng.integers(0, 30, 40), rng.integers(0, 20, 40)], 1).astype(np.float32)
    cases = {"v2": ({"coords": grid * 256.0}, {"tile_size": 256.0, "unit": "um"}),
             "current": ({"coords": grid * 112.5}, {"tile_size_um": 112.5, "tile_size_px": 224, "unit": "um", "stamp_version": "2.4.0"}),
             "historic_attr": ({"coords": grid * 224.0 + 7.0}, {"tile_size": 224}),
             "historic_stride": ({"coords": grid * 224.0}, {}),
             "no_coords": ({"patch_embeddings": np.zeros((13, 8), np.float32)}, {}),
             "newer": ({"coords": grid * 256.0}, {"tile_size_um": 256.0, "stamp_version": "9.1.0"}),
             "unknown": ({"coords": grid * 300.0}, {})}
    out = {}
    for name, (ds, attrs) in cases.items():
        rec = {"datasets": {k: v.tolist() for k, v in ds.items()}, "attrs": attrs}
        try:
            ci = glb["get_coords"](FileStub(ds, attrs))
            rec.update(coords_um=np.asarray(ci.coords_um).tolist(), tile_size_um=float(ci.tile_size_um), tile_size_px=None if ci.tile_size_px is None else int(ci.tile_size_px))
        except Exception as e:  # noqa: BLE001
            rec.update(error=type(e).__name__, message=str(e))
        out[name] = rec
    OUT.mkdir(parents=True, exist_ok=True)
    (OUT / "get_coords.json").write_text(json.dumps(out))
    print("wrote get_coords.json", {k: v.get("error", "ok") for k, v in out.items()})


def golden_mil_vit() -> None:
    vt = load_by_path("stamp.modeling.models.vision_tranformer", REF / "modeling" / "models" / "vision_tranformer.py")
    for tag, use_alibi, kw in (
        ("plain", False, dict(dim_output=3, dim_input=48, dim_model=64, n_layers=2, n_heads=2, dim_feedforward=96)),
        ("alibi", True, dict(dim_output=2, dim_input=40, dim_model=64, n_layers=2, n_heads=4, dim_feedforward=64)),
    ):
        torch.manual_seed(7 if use_alibi else 6)
        model = vt.VisionTransformer(dropout=0.0, use_alibi=use_alibi, **kw)
        bags = torch.randn(3, 37, kw["dim_input"])
        coords = torch.rand(3, 37, 2) * 4000
        mask = torch.zeros(3, 37, dtype=torch.bool)
        mask[1, 30:] = True
        mask[2, 11:] = True
        arrs = dict(bags=bags.numpy(), coords=coords.numpy(), mask=mask.numpy(), **sd_np(model))
        if use_alibi:
            # a few train-mode forwards update the running-mean buffers exactly as training would
            model.train()
            for i in range(2):
                with torch.no_grad():
                    model(bags + i, coords=coords * (1 + i), mask=None)
            arrs.update({k.replace("w:", "w_after_train:"): v for k, v in sd_np(model).items() if "running_mean" in k or "items_so_far" in k})
            arrs.update(sd_np(model))          # eval goldens use the post-update buffers
        model.eval()
        with torch.no_grad():
            arrs["logits_nomask"] = model(bags, coords=coords, mask=None).numpy()
            arrs["logits_mask"] = model(bags, coords=coords, mask=mask).numpy()
        arrs["hparams"] = np.array([kw[k] for k in ("dim_output", "dim_input", "dim_model", "n_layers", "n_heads", "dim_feedforward")])
        save(f"mil_vit_{tag}.npz", **arrs)


def golden_mil_vit_train() -> None:
    """TRAIN-mode fixtures: one forward + backward of the reference module in .train(), with the keep masks its own nn.Dropout
    modules drew captured by forward hooks (project_features.2, transformer.layers.{l}.1.3 and .1.5).  nn.MultiheadAttention's
    internal dropout cannot be observed from outside, so the plain variant uses dropout=0.0 (its feed-forward Dropouts are live
    regardless: hard-coded 0.5) and the ALiBi variant -- which has no attention dropout -- uses dropout=0.3."""
    vt = sys.modules.get("stamp.modeling.models.vision_tranformer") or load_by_path(
        "stamp.modeling.models.vision_tranformer", REF / "modeling" / "models" / "vision_tranformer.py")
    for tag, use_alibi, p_drop, kw in (
        ("plain", False, 0.0, dict(dim_output=3, dim_input=48, dim_model=64, n_layers=2, n_heads=2, dim_feedforward=96)),
        ("alibi", True, 0.3, dict(dim_output=2, dim_input=40, dim_model=64, n_layers=2, n_heads=4, dim_feedforward=64)),
    ):
        torch.manual_seed(17 if use_alibi else 16)
        model = vt.VisionTransformer(dropout=p_drop, use_alibi=use_alibi, **kw).train()
        bags = torch.randn(3, 37, kw["dim_input"])
        coords = torch.rand(3, 37, 2) * 4000
        targets = torch.nn.functional.one_hot(torch.tensor([0, 1, 1]), kw["dim_output"]).float()
        weights = torch.rand(kw["dim_output"]) + 0.5
        arrs = dict(bags=bags.numpy(), coords=coords.numpy(), targets=targets.numpy(), class_weights=weights.numpy(),
                    **sd_np(model, "w_before:"))
        masks, rates = {}, {}

        def hook(name):
            def fn(mod, inp, out):
                masks[name] = ((out != 0) | (inp[0] == 0)).to(torch.uint8).numpy()
                rates[name] = mod.p
            return fn
        handles = [m.register_forward_hook(hook(n)) for n, m in model.named_modules() if isinstance(m, torch.nn.Dropout)]
        logits = model(bags, coords=coords, mask=None)
        loss = torch.nn.functional.cross_entropy(logits, targets, weight=weights)      # LitTileClassifier._step, models/__init__.py:254-258
        loss.backward()
        for h in handles:
            h.remove()
        assert set(masks) == {"project_features.2"} | {f"transformer.layers.{l}.1.{i}" for l in range(kw["n_layers"]) for i in (3, 5)}, set(masks)
        arrs.update({"mask:" + k: v for k, v in masks.items()})
        arrs.update({"rate:" + k: np.float32(v) for k, v in rates.items()})
        arrs["logits"], arrs["loss"] = logits.detach().numpy(), loss.detach().numpy()
        arrs.update({"g:" + n: p.grad.numpy() for n, p in model.named_parameters()})
        arrs.update(sd_np(model, "w_after:"))            # ALiBi: scaler buffers after the train-mode forward
        arrs["hparams"] = np.array([kw[k] for k in ("dim_output", "dim_input", "dim_model", "n_layers", "n_heads", "dim_feedforward")])
        arrs["dropout"] = np.float32(p_drop)
        save(f"mil_vit_train_{tag}.npz", **arrs)


def golden_transmil() -> None:
    tm = load_by_path("stamp.modeling.models.trans_mil", REF / "modeling" / "models" / "trans_mil.py")
    for tag, T, dim_in, dim_h in (("t50", 50, 24, 64), ("t300", 300, 32, 64)):
        torch.manual_seed(11 + T)
        model = tm.TransMIL(dim_output=2, dim_input=dim_in, dim_hidden=dim_h).eval()
        bags = torch.randn(2, T, dim_in)
        with torch.no_grad():
            logits = model(bags)
            # intermediates for kernel-level checks
            x = torch.randn(2, 70, dim_h)
            attn_out = model.layer1.attn(x)
            a2 = torch.softmax(torch.randn(2, 8, 32, 32), dim=-1)
            pinv = tm.moore_penrose_iter_pinv(a2, 6)
            ppeg = model.pos_layer(torch.randn(2, 1 + 64, dim_h), 8, 8)
        save(f"transmil_{tag}.npz", bags=bags.numpy(), logits=logits.numpy(), nys_x=x.numpy(), nys_out=attn_out.numpy(),
             pinv_in=a2.numpy(), pinv_out=pinv.numpy(), hparams=np.array([2, dim_in, dim_h]), **sd_np(model))
    # PPEG input for replay
    torch.manual_seed(5)
    model = tm.TransMIL(dim_output=2, dim_input=8, dim_hidden=64).eval()
    xin = torch.randn(2, 1 + 64, 64)
    with torch.no_grad():
        out = model.pos_layer(xin, 8, 8)
    save("transmil_ppeg.npz", x=xin.numpy(), out=out.numpy(), **sd_np(model.pos_layer))


def golden_mlp_cox_transforms() -> None:
    mlp = load_by_path("stamp.modeling.models.mlp", REF / "modeling" / "models" / "mlp.py")
    torch.manual_seed(3)
    m = mlp.MLP(dim_input=20, dim_hidden=16, dim_output=3, num_layers=3, dropout=0.0).eval()
    lin = mlp.Linear(dim_input=20, dim_output=2).eval()
    x3, x2 = torch.randn(4, 9, 20), torch.randn(4, 20)
    with torch.no_grad():
        save("mlp.npz", x3=x3.numpy(), x2=x2.numpy(), mlp_y3=m(x3).numpy(), mlp_y2=m(x2).numpy(),
             lin_y3=lin(x3).numpy(), lin_y2=lin(x2).numpy(),
             **sd_np(m, "mlp:"), **sd_np(lin, "lin:"))
    cox = load_by_path("stamp.modeling.models.cox", REF / "modeling" / "models" / "cox.py")
    torch.manual_seed(4)
    cases = {}
    # the reference's own docstring known answers (cox.py:192-204)
    log_hz = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5])
    event = torch.tensor([1, 0, 1, 0, 1], dtype=torch.bool)
    time = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0])
    cases["doc_mean"] = cox.neg_partial_log_likelihood(log_hz, time, event).item()
    cases["doc_sum"] = cox.neg_partial_log_likelihood(log_hz, time, event, reduction="sum").item()
    time_t = torch.tensor([1.0, 2.0, 2.0, 4.0, 5.0])
    cases["doc_tie_efron"] = cox.neg_partial_log_likelihood(log_hz, time_t, event, ties_method="efron").item()
    cases["doc_tie_breslow"] = cox.neg_partial_log_likelihood(log_hz, time_t, event, ties_method="breslow").item()
    lh = torch.randn(40)
    tt = torch.randint(1, 15, (40,)).float()          # many ties
    ev = torch.rand(40) < 0.7
    save("cox.npz", log_hz=lh.numpy(), time=tt.numpy(), event=ev.numpy(),
         efron=np.array(cox.neg_partial_log_likelihood(lh, tt, ev, ties_method="efron").item()),
         breslow=np.array(cox.neg_partial_log_likelihood(lh, tt, ev, ties_method="breslow").item()),
         notie=np.array(cox.neg_partial_log_likelihood(lh, torch.arange(40.0), ev).item()),
         **{k: np.array(v) for k, v in cases.items()})
    # the slide / patient-level survival objective: Breslow `cox_loss`, a static method of the Lit class (models/__init__.py:625-659)
    cox_loss = exec_method(REF / "modeling" / "models" / "__init__.py", "cox_loss", {"torch": torch})
    torch.manual_seed(6)
    sc = torch.randn(64, 1, requires_grad=True)
    tm_ = torch.randint(30, 2000, (64,)).float()
    tm_[10:14] = tm_[9]                                  # a few ties
    evn = (torch.rand(64) < 0.7).float()
    l_ = cox_loss(sc, tm_, evn)
    g_, = torch.autograd.grad(l_, sc)
    sc0 = torch.randn(5, requires_grad=True)
    l0 = cox_loss(sc0, torch.arange(5.0), torch.zeros(5))           # no events: `scores.sum() * 0.0` (keeps the graph)
    save("cox_slide.npz", scores=sc.detach().numpy(), times=tm_.numpy(), events=evn.numpy(), loss=np.array(l_.item()), grad=g_.numpy(),
         loss_no_event=np.array(l0.item()), no_event_requires_grad=np.array(bool(l0.requires_grad)))
    tr = load_by_path("stamp.modeling.transforms", REF / "modeling" / "transforms.py")
    out = {}
    for dt, name in ((torch.float32, "f32"), (torch.float16, "f16"), (torch.bfloat16, "bf16")):
        torch.manual_seed(21)
        data = (torch.randn(64, 33) * 100).to(dt)
        torch.manual_seed(22)      # the reference draws torch.randint from the global CPU generator
        aug = tr.vary_precision(data, min_fraction_bits=2)
        out[f"in_{name}"] = data.view(torch.int16 if dt != torch.float32 else torch.int32).numpy()
        out[f"out_{name}"] = aug.view(torch.int16 if dt != torch.float32 else torch.int32).numpy()
    save("vary_precision.npz", **out)


def golden_bag() -> None:
    # modeling/data.py imports h5py at module top; exec only the pure-torch function
    glb = {"torch": torch, "_Bag": typing.Any, "_Coordinates": typing.Any, "BagSize": int}
    exec_defs(REF / "modeling" / "data.py", {"_to_fixed_size_bag"}, glb)
    f = glb["_to_fixed_size_bag"]
    out = {}
    for n, bs in ((10, 16), (100, 16), (16, 16), (1000, 512), (1, 4)):
        torch.manual_seed(n)
        bag = torch.randn(n, 6)
        coords = torch.rand(n, 2)
        torch.manual_seed(1234)
        b1, c1, l1 = f(bag, coords, bs, deterministic=False)
        b2, c2, l2 = f(bag, coords, bs, deterministic=True)
        out.update({f"bag_{n}_{bs}": bag.numpy(), f"coords_{n}_{bs}": coords.numpy(),
                    f"rand_bag_{n}_{bs}": b1.numpy(), f"rand_coords_{n}_{bs}": c1.numpy(), f"rand_len_{n}_{bs}": np.array(l1),
                    f"det_bag_{n}_{bs}": b2.numpy(), f"det_coords_{n}_{bs}": c2.numpy(), f"det_len_{n}_{bs}": np.array(l2)})
    save("fixed_size_bag.npz", **out)



def golden_bag_dataset() -> None:
    """The reference's bag-building HOST layer, executed as it stands (src/stamp/modeling/data.py): `BagDataset.__getitem__` (:584-655: patient -> slide
    files concatenated, `.float()`, transform, `_to_fixed_size_bag`), `_collate_to_tuple` / `_collate_multitarget` (:255-295), `_parse_targets` (:146-252)
    and `_compute_class_weights_and_check_categories` (modeling/train.py:567-621).  data.py imports h5py at module top: `h5py.File(path, ...)` is a
    stand-in that serves in-memory datasets / attributes (the same objects `golden_get_coords` hands to `get_coords`).  The fixture holds the files'
    contents, the calls' arguments and what the reference returned -- tests/test_cpu_bags.py replays them through stamp_amd.bags over REAL .h5 files."""
    import json
    import logging
    from collections import OrderedDict
    from collections.abc import Callable, Iterable, Mapping, Sequence
    from dataclasses import KW_ONLY, dataclass
    from typing import Any, Dict, Generic, List, TypeVar, Union, cast

    from packaging.version import Version
    from torch.utils.data import DataLoader, Dataset

    class DS:                                                        # h5py.Dataset stand-in
        def __init__(self, arr):
            self.arr = np.asarray(arr)
            self.shape = self.arr.shape

        def __getitem__(self, k):
            return self.arr[k]

    registry: dict = {}
    opened: list = []

    class FileStub(dict):
        def __init__(self, path, mode="r", **kw):
            ds, attrs = registry[str(path)]
            super().__init__({k: DS(v) for k, v in ds.items()})
            self.attrs, self.filename = dict(attrs), str(path)
            opened.append(str(path))

        def close(self):
            pass

    h5 = types.SimpleNamespace(Dataset=DS, File=FileStub)
    T = torch.Tensor
    glb = {"np": np, "torch": torch, "h5py": h5, "dataclass": dataclass, "KW_ONLY": KW_ONLY, "Version": Version, "stamp": types.SimpleNamespace(__version__="2.5.0"),
           "cast": cast, "Tensor": T, "Microns": float, "TilePixels": int, "SlideMPP": float, "_logger": logging.getLogger("golden"), "Dataset": Dataset,
           "DataLoader": DataLoader, "OrderedDict": OrderedDict, "Sequence": Sequence, "Iterable": Iterable, "Callable": Callable, "Mapping": Mapping, "Any": Any,
           "Dict": Dict, "List": List, "Union": Union, "Generic": Generic, "FeaturePath": Path, "_BinaryIOLike": Any, "BagSize": int, "_Bag": T, "_Coordinates": T,
           "_EncodedTarget": Any, "Bags": T, "CoordinatesBatch": T, "BagSizes": T, "EncodedTargets": T, "Category": str, "Task": str, "GroundTruthType": TypeVar("G"),
           "PatientFeatureDataset": type("PatientFeatureDataset", (), {})}
    exec_defs(REF / "modeling" / "data.py", {"CoordsInfo", "get_coords", "get_stride", "_to_fixed_size_bag", "BagDataset", "_collate_to_tuple", "_collate_multitarget",
                                               "_parse_targets", "PatientData"}, glb, inherit_future=False)
    exec_defs(REF / "modeling" / "train.py", {"_compute_class_weights_and_check_categories"}, glb)
    rng = np.random.default_rng(11)
    F = 12

    def slide(n, fmt, dtype=np.float16):
        grid = np.stack([rng.integers(0, 40, n), rng.integers(0, 30, n)], 1).astype(np.float32)
        feats = rng.standard_normal((n, F)).astype(dtype)
        if fmt == "current":
            return {"feats": feats, "coords": grid * 256.0}, {"tile_size_um": 256.0, "tile_size_px": 224, "unit": "um", "stamp_version": "2.4.0"}
        if fmt == "v2":
            return {"feats": feats, "coords": grid * 128.0}, {"tile_size": 128.0, "unit": "um"}
        if fmt == "historic":
            return {"feats": feats, "coords": grid * 224.0}, {}
        return {"patch_embeddings": feats}, {}                      # coords-less bypass (:743-757)
    files = {"a1": slide(37, "current"), "a2": slide(5, "current"), "b1": slide(70, "v2", np.float32), "c1": slide(9, "historic"), "c2": slide(11, "historic"),
             "c3": slide(3, "historic"), "d1": slide(20, "nocoords"), "e1": slide(64, "current")}
    for k, (ds, attrs) in files.items():
        registry[f"/feat/{k}.h5"] = (ds, attrs)
    bags = [[Path("/feat/a1.h5"), Path("/feat/a2.h5")], [Path("/feat/b1.h5")], [Path("/feat/c1.h5"), Path("/feat/c2.h5"), Path("/feat/c3.h5")], [Path("/feat/d1.h5")],
            [Path("/feat/e1.h5")]]
    out: dict = {"files": {k: {"datasets": {n: {"dtype": str(v.dtype), "data": v.tolist()} for n, v in ds.items()}, "attrs": attrs} for k, (ds, attrs) in files.items()},
                 "bags": [[p.stem for p in b] for b in bags], "cases": {}}
    # ---- targets as the reference encodes them
    PD = glb["PatientData"]
    gts_cls = ["lum", "basal", "lum", "her2", "basal"]
    pdata = [PD(ground_truth=g, feature_files=b) for g, b in zip(gts_cls, bags)]
    y_cls, cats = glb["_parse_targets"](patient_data=pdata, task="classification", categories=None)
    y_cls_fixed, cats_fixed = glb["_parse_targets"](patient_data=pdata, task="classification", categories=["her2", "lum", "basal", "normal"])
    y_reg, _ = glb["_parse_targets"](patient_data=[PD(ground_truth=g, feature_files=b) for g, b in zip([1.5, None, -2.0, 0.25, 7.0], bags)], task="regression")
    y_surv, _ = glb["_parse_targets"](patient_data=[PD(ground_truth=g, feature_files=b) for g, b in zip([(10.0, 1), (3.5, 0), None, ("nan", 1), (7, None)], bags)],
                                      task="survival")
    gts_multi = [{"sub": "lum", "grade": "g1"}, {"sub": "basal", "grade": None}, None, {"sub": "her2", "grade": "g3"}, {"sub": "lum", "grade": "g1"}]
    y_multi, cats_multi = glb["_parse_targets"](patient_data=[PD(ground_truth=g, feature_files=b) for g, b in zip(gts_multi, bags)], task="classification")
    out["targets"] = {"classification": {"gts": gts_cls, "encoded": y_cls.tolist(), "categories": list(cats)},
                      "classification_fixed": {"categories_in": ["her2", "lum", "basal", "normal"], "encoded": y_cls_fixed.tolist(), "categories": list(cats_fixed)},
                      "regression": {"gts": [1.5, None, -2.0, 0.25, 7.0], "encoded": [[None if np.isnan(v) else v for v in r] for r in y_reg.tolist()]},
                      "survival": {"gts": [[10.0, 1], [3.5, 0], None, ["nan", 1], [7, None]], "encoded": [[None if np.isnan(v) else v for v in r] for r in y_surv.tolist()]},
                      "multi": {"gts": gts_multi, "encoded": [{k: v.tolist() for k, v in d.items()} for d in y_multi], "categories": {k: list(v) for k, v in cats_multi.items()}}}
    try:
        glb["_parse_targets"](patient_data=[PD(ground_truth="x", feature_files=b) for b in bags], task="classification")
    except ValueError as e:
        out["targets"]["one_class_error"] = str(e)
    # ---- items and batches
    BD = glb["BagDataset"]

    def item_rec(it):
        bag, coords, n, tgt = it
        return {"bag": bag.tolist(), "bag_dtype": str(bag.dtype), "coords": coords.tolist(), "n": int(n),
                "target": ({k: v.tolist() for k, v in tgt.items()} if isinstance(tgt, dict) else torch.as_tensor(tgt).tolist())}
    for name, kw, seed in (("det_16", dict(bag_size=16, deterministic=True), None), ("rand_16", dict(bag_size=16, deterministic=False), 99),
                           ("rand_40", dict(bag_size=40, deterministic=False), 7), ("all", dict(bag_size=None), None)):
        ds = BD(bags=bags, ground_truths=y_cls, transform=None, **kw)
        if seed is not None:
            torch.manual_seed(seed)
        items = [ds[i] for i in range(len(bags))]
        rec = {"kw": {k: v for k, v in kw.items()}, "seed": seed, "items": [item_rec(it) for it in items]}
        if kw["bag_size"] is not None:
            b, c, s, t = glb["_collate_to_tuple"](items)
            rec["batch"] = {"bags_shape": list(b.shape), "coords_shape": list(c.shape), "bag_sizes": s.tolist(), "bag_sizes_dtype": str(s.dtype), "targets": t.tolist()}
        out["cases"][name] = rec
    # a transform, multi-target ground truths, scalar / 2-d targets through the collate's shape rule
    ds = BD(bags=bags, bag_size=8, ground_truths=y_multi, transform=lambda x: x * 2.0 + 1.0, deterministic=True)
    items = [ds[i] for i in range(len(bags))]
    b, c, s, t = glb["_collate_multitarget"](items)
    out["cases"]["multi_det_8_transform"] = {"items": [item_rec(it) for it in items], "batch": {"bags_shape": list(b.shape), "bag_sizes": s.tolist(),
                                                                                             "targets": {k: v.tolist() for k, v in t.items()}}}
    ds = BD(bags=bags, bag_size=4, ground_truths=y_surv, transform=None, deterministic=True)
    items = [ds[i] for i in range(len(bags))]
    fake = [(items[0][0], items[0][1], items[0][2], torch.tensor(3.0)), (items[1][0], items[1][1], items[1][2], torch.tensor([[1.0, 2.0]]).view(1, 2)[0, :1])]
    _, _, _, tt = glb["_collate_to_tuple"](fake)
    out["cases"]["collate_shapes"] = {"targets_in": [3.0, [1.0]], "targets_out": tt.tolist()}
    # the handle cache: at most 128 open files, least recently used closed first (:596-612)
    for i in range(140):
        registry[f"/feat/many{i}.h5"] = files["a2"]
    ds = BD(bags=[[Path(f"/feat/many{i}.h5")] for i in range(140)], bag_size=2, ground_truths=torch.zeros(140, 2), transform=None, deterministic=True)
    opened.clear()
    for i in list(range(140)) + [0, 139, 11]:
        ds[i]
    out["cases"]["handle_cache"] = {"opens": len(opened), "cached_after": len(ds._h5_handle_cache), "reopened": [Path(p).stem for p in opened[140:]]}
    # ---- inverse-frequency class weights (train.py:567-621)
    f = glb["_compute_class_weights_and_check_categories"]
    big = torch.zeros(60, 3)
    big[:30, 0] = 1
    big[30:50, 1] = 1
    big[50:, 2] = 1
    dl = types.SimpleNamespace(dataset=BD(bags=[[]] * 60, ground_truths=big, transform=None))
    w = f(train_dl=dl, feature_type="tile", train_categories=["a", "b", "c"])
    dlm = types.SimpleNamespace(dataset=BD(bags=bags, ground_truths=y_multi, transform=None))
    wm = f(train_dl=dlm, feature_type="tile", train_categories=cats_multi)
    out["class_weights"] = {"single": {"ground_truths": big.tolist(), "weights": w.tolist()}, "multi": {k: [None if np.isnan(x) or np.isinf(x) else x for x in v.tolist()] for k, v in wm.items()},
                            "multi_raw": {k: [repr(float(x)) for x in v.tolist()] for k, v in wm.items()}}
    try:
        f(train_dl=dl, feature_type="tile", train_categories=["only"])
    except ValueError as e:
        out["class_weights"]["one_category_error"] = str(e)
    OUT.mkdir(parents=True, exist_ok=True)
    (OUT / "bag_dataset.json").write_text(json.dumps(out))
    print("wrote bag_dataset.json", (OUT / "bag_dataset.json").stat().st_size // 1024, "KiB")


from oracle.tiling import synthetic_slide  # noqa: E402  (seeded generator shared with the tests)


class FakeSlide:
    """The slice of openslide's AbstractSlide the reference's tiling code touches: `dimensions`, `read_region` (RGBA, transparent past
    the edge like openslide), `get_thumbnail` (white background + PIL thumbnail, as openslide-python implements it)."""

    def __init__(self, rgb: np.ndarray, mpp: float):
        from PIL import Image
        self._im = Image.fromarray(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]), "RGBA")
        self.dimensions = (rgb.shape[1], rgb.shape[0])
        self.properties = {"openslide.mpp-x": str(mpp)}

    def read_region(self, location, level, size):
        from PIL import Image
        assert level == 0
        x, y = location
        out = Image.new("RGBA", size, (0, 0, 0, 0))
        out.paste(self._im.crop((x, y, min(x + size[0], self.dimensions[0]), min(y + size[1], self.dimensions[1]))), (0, 0))
        return out

    def get_thumbnail(self, size):
        from PIL import Image
        bg = Image.new("RGB", self._im.size, "#ffffff")
        thumb = Image.composite(self._im, bg, self._im)
        thumb.thumbnail(tuple(int(v) for v in size), Image.Resampling.LANCZOS)
        return thumb


def golden_tiling() -> None:
    """The reference's own `_foreground_coords`, `_supertiles` and `_tiles` (src/stamp/preprocessing/tiling.py:196-347) run on a synthetic
    slide object.  Stored: the thumbnail the fake slide handed out, the foreground origins, every tile's coordinates and CRC32, and a
    few tiles in full (an edge supertile that reads past the slide is among them)."""
    import zlib
    from concurrent import futures
    from dataclasses import dataclass
    from typing import Generic, NamedTuple, TypeVar, cast

    from PIL import Image

    glb = {"np": np, "futures": futures, "Image": Image, "cast": cast, "Iterator": typing.Iterator, "NamedTuple": NamedTuple, "Generic": Generic,
           "TypeVar": TypeVar, "dataclass": dataclass, "_Unit": TypeVar("_Unit"), "Microns": float, "TilePixels": int, "SlidePixels": int, "SlideMPP": float,
           "npt": types.SimpleNamespace(NDArray=typing.Any), "openslide": types.SimpleNamespace(AbstractSlide=object),
           "get_slide_mpp_": lambda slide, default_mpp=None: float(slide.properties["openslide.mpp-x"])}
    # `class _Tile(NamedTuple, Generic[_Unit])` needs Python >= 3.11: arithmetic-free stand-ins for the two record types (tiling.py:51-65)
    import collections

    class _XYCoords:
        def __init__(self, x, y):
            self.x, self.y = x, y

        def __class_getitem__(cls, item):
            return cls

    class _Tile(collections.namedtuple("_Tile", "image coordinates size")):
        def __class_getitem__(cls, item):
            return cls

    glb.update(_XYCoords=_XYCoords, _Tile=_Tile)
    exec_defs(REF / "preprocessing" / "tiling.py", {"_tiles", "_foreground_coords", "_supertiles"}, glb)
    for tag, (w, h, mpp, seed) in {"mpp050": (3000, 2100, 0.5, 11), "mpp025": (2500, 1900, 0.25, 12)}.items():
        slide = FakeSlide(synthetic_slide(w, h, seed), mpp)
        kw = dict(tile_size_um=256.0, tile_size_px=224, max_supertile_size_slide_px=1024, max_workers=1, brightness_cutoff=224, default_slide_mpp=None)
        ts_px = int(np.ceil(256.0 / mpp)) * max(int(1024 * mpp // 256.0), 1)
        fg = [(c.x, c.y) for c in glb["_foreground_coords"](slide, ts_px, 224)]
        grid = np.ceil(np.array(slide.dimensions) / ts_px).astype(np.uint32)
        thumb = np.array(slide.get_thumbnail(tuple(grid * 2)))
        tiles = list(glb["_tiles"](slide=slide, **kw))
        tiles.sort(key=lambda t: (t.coordinates.y, t.coordinates.x))
        coords = np.array([(t.coordinates.x, t.coordinates.y) for t in tiles], dtype=np.float64)
        crc = np.array([zlib.crc32(np.array(t.image).tobytes()) for t in tiles], dtype=np.uint32)
        keep = sorted({0, len(tiles) // 2, len(tiles) - 1})
        full = np.stack([np.array(tiles[i].image) for i in keep])
        save(f"tiling_{tag}.npz", slide=np.array([w, h, seed]), mpp=np.float64(mpp), thumb2x=thumb, foreground=np.array(fg, dtype=np.int64),
             coords_um=coords, crc32=crc, full_idx=np.array(keep), full_tiles=full)


def golden_tile_cache() -> None:
    """Tile-cache zips (tiling.py:68-168, 380-406).  `tile_cache_amd.zip` is written by stamp_amd.tile_cache and must be readable by
    the REFERENCE's `_tiles_from_cache_file`; `tile_cache_ref_style.zip` is assembled with the reference writer's own statements
    (:121-151: json entry, f-string entry names, PIL save) in the legacy form without `tile_ext` and is what stamp_amd must read."""
    import collections
    import io
    import json
    import re
    from zipfile import ZipFile

    from PIL import Image

    from stamp_amd import tile_cache as tc

    class _XYCoords:
        def __init__(self, x, y):
            self.x, self.y = x, y

    _Tile = collections.namedtuple("_Tile", "image coordinates size")
    glb = {"ZipFile": ZipFile, "json": json, "re": re, "Image": Image, "Path": Path, "Iterator": typing.Iterator, "_Tile": _Tile, "_XYCoords": _XYCoords,
           "_TilerParams": dict, "Microns": float}
    exec_defs(REF / "preprocessing" / "tiling.py", {"_tiles_from_cache_file"}, glb)
    rng = np.random.default_rng(21)
    tiles = rng.integers(0, 256, (5, 32, 32, 3), dtype=np.uint8)
    coords = np.array([[0.0, 0.0], [256.0, 0.0], [512.5, 1024.0], [0.0, 1280.0], [123456.0, 7.25]])
    params = tc.tiler_params("/data/slide_x.svs", tile_size_um=256.0, tile_size_px=32, max_supertile_size_slide_px=1024, brightness_cutoff=224,
                             code_sha256="0" * 64, tile_ext="png")
    OUT.mkdir(parents=True, exist_ok=True)
    tc.write_tile_cache(OUT / "tile_cache_amd.zip", tiles, coords, params)
    got = list(glb["_tiles_from_cache_file"](OUT / "tile_cache_amd.zip"))             # the reference reads what stamp_amd wrote
    assert len(got) == 5 and all(np.array_equal(np.array(t.image), tiles[i]) for i, t in enumerate(got))
    assert [(t.coordinates.x, t.coordinates.y) for t in got] == [tuple(c) for c in coords.tolist()] and got[0].size == 256.0
    legacy = {k: v for k, v in params.items() if k != "tile_ext"}
    with ZipFile(OUT / "tile_cache_ref_style.zip", "w") as zf:                           # the reference writer's statements, jpg, no tile_ext
        with zf.open("tiler_params.json", "w") as fp:
            fp.write(json.dumps(legacy).encode())
        for t, (x, y) in zip(tiles, coords):
            with zf.open(f"tile_({float(x)}, {float(y)}).jpg", "w") as fp:
                Image.fromarray(t, "RGB").save(fp, format="jpeg")
    ref = list(glb["_tiles_from_cache_file"](OUT / "tile_cache_ref_style.zip"))
    save("tile_cache_expect.npz", tiles=tiles, coords=coords, jpeg_decoded=np.stack([np.array(t.image) for t in ref]))


def he_like_tiles(n: int, size: int, seed: int) -> np.ndarray:
    """Synthetic H&E-looking u8 tiles (smooth mixtures of two stain colours on white) -- smoother statistics than
    uniform noise, so the conv stem and the shifted-window masks see structured input."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    out = np.empty((n, size, size, 3), np.uint8)
    hema, eos = np.array([0.65, 0.70, 0.29], np.float32), np.array([0.07, 0.99, 0.11], np.float32)
    for i in range(n):
        f = rng.uniform(1.0, 6.0, size=4)
        ph = rng.uniform(0, 6.28, size=4)
        a = 0.5 + 0.5 * np.sin(6.28 * f[0] * xx + ph[0]) * np.cos(6.28 * f[1] * yy + ph[1])
        b = 0.5 + 0.5 * np.sin(6.28 * f[2] * (xx + yy) + ph[2]) * np.cos(6.28 * f[3] * (xx - yy) + ph[3])
        od = a[..., None] * 1.2 * hema + b[..., None] * 0.8 * eos
        img = 255.0 * np.exp(-od) + rng.normal(0, 4.0, size=od.shape)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def golden_ctranspath() -> None:
    """Reference `_SwinTransformer` + `_ConvStem` (preprocessing/extractor/ctranspath.py) on seeded weights.
    The module top imports gdown / torchvision / stamp.utils.cache; only its model definitions are executed."""
    import math
    import warnings
    from collections.abc import Iterable
    from itertools import repeat
    from typing import Optional, TypeVar, cast

    import torch.nn as nn
    import torch.utils.checkpoint as checkpoint
    from torch import _assert
    from torch.nn.init import _calculate_fan_in_and_fan_out

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from stamp_amd.swin import SWIN_PRESETS, random_swin_state_dict

    glb = {"torch": torch, "nn": nn, "math": math, "warnings": warnings, "Iterable": Iterable, "repeat": repeat,
           "Optional": Optional, "TypeVar": TypeVar, "cast": cast, "checkpoint": checkpoint, "_assert": _assert,
           "_calculate_fan_in_and_fan_out": _calculate_fan_in_and_fan_out, "Tensor": torch.Tensor,
           "_T": TypeVar("_T"), "__name__": "ref_ctranspath"}
    names = {"_to_2tuple", "_no_grad_trunc_normal_", "_trunc_normal_tf_", "_variance_scaling_", "_trunc_normal_",
             "_lecun_normal_", "_init_vit_weights", "_window_partition", "_window_reverse", "_drop_path", "_DropPath",
             "_PatchEmbed", "_Mlp", "_ConvStem", "_WindowAttention", "_SwinTransformerBlock", "_PatchMerging",
             "_BasicLayer", "_SwinTransformer", "_swin_tiny_patch4_window7_224"}
    exec_defs(REF / "preprocessing" / "extractor" / "ctranspath.py", names, glb)

    def run(tag: str, model: torch.nn.Module, cfg, seed: int, n: int, tap_rows: int):
        model.head = nn.Identity()                       # ctranspath.py:51
        sd = random_swin_state_dict(cfg, seed)
        ref_sd = model.state_dict()
        float_keys = {k for k, v in ref_sd.items() if v.is_floating_point() and not k.endswith("attn_mask")}
        assert float_keys == set(sd), (float_keys ^ set(sd))
        for k in float_keys:
            assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
        model.load_state_dict({**ref_sd, **sd}, strict=True)     # buffers (rel-pos index, attn_mask, counters) stay
        model.eval()
        rng = np.random.default_rng(seed + 77)
        tiles = np.concatenate([rng.integers(0, 256, size=(n - n // 2, cfg.img, cfg.img, 3), dtype=np.uint8),
                                he_like_tiles(n // 2, cfg.img, seed + 78)])
        x = torch.from_numpy(tiles).permute(0, 3, 1, 2).float() / 255.0
        x = (x - torch.tensor(cfg.mean).view(1, 3, 1, 1)) / torch.tensor(cfg.std).view(1, 3, 1, 1)   # ctranspath.py:56-64
        taps = {}
        with torch.no_grad():
            t = model.patch_embed(x)
            taps["stem"] = t
            for s, layer in enumerate(model.layers):
                t = layer(t)
                taps[f"stage{s}"] = t
            feats = model(x)
        arrs = {"tiles": tiles, "feats": feats.numpy(), "seed": np.array(seed)}
        for k, v in taps.items():      # a strided sample of token rows keeps the fixture small
            L = v.shape[1]
            idx = np.unique(np.linspace(0, L - 1, min(L, tap_rows)).round().astype(np.int64))
            arrs["tap_idx_" + k] = idx
            arrs["tap_" + k] = v[:, idx].numpy()
        save(f"ctranspath_{tag}.npz", **arrs)

    full = SWIN_PRESETS["ctranspath"]
    run("swin_t", glb["_swin_tiny_patch4_window7_224"](embed_layer=glb["_ConvStem"], pretrained=False), full, 0, 4, 40)
    tiny = SWIN_PRESETS["test_swin_tiny"]
    run("tiny", glb["_SwinTransformer"](img_size=tiny.img, embed_dim=tiny.embed, depths=tiny.depths, num_heads=tiny.heads,
                                        window_size=7, embed_layer=glb["_ConvStem"]), tiny, 1, 6, 64)


def golden_texture_gray() -> None:
    """`tile.convert("L")` of the reference's texture filter (tiling.py:284), executed with the installed Pillow."""
    from PIL import Image

    rng = np.random.default_rng(5)
    tiles = np.concatenate([rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8), he_like_tiles(1, 224, 6),
                            np.stack([np.stack(np.meshgrid(np.arange(224) % 256, (np.arange(224) * 3) % 256), -1).astype(np.uint8)[..., [0, 1, 0]]])])
    tiles[2, ..., 2] = 255 - tiles[2, ..., 0]
    gray = np.stack([np.array(Image.fromarray(t).convert("L")) for t in tiles])
    save("texture_gray.npz", tiles=tiles, gray=gray)


def golden_slide_mpp() -> None:
    """The reference's `get_slide_mpp_` / `_extract_mpp_from_comments` / `_extract_mpp_from_metadata` (src/stamp/preprocessing/tiling.py:409-475) on slide
    stand-ins that carry only a `properties` mapping (the one thing the functions read; openslide itself is absent: the module constant
    PROPERTY_NAME_MPP_X is handed in with openslide-python's value).  Cases: every source, their precedence, malformed XML, nothing at all +- a default."""
    import json
    import logging
    import re
    from xml.dom import minidom

    class MPPExtractionError(Exception):
        pass

    glb = {"openslide": types.SimpleNamespace(PROPERTY_NAME_MPP_X="openslide.mpp-x", AbstractSlide=object, open_slide=None), "Path": Path, "SlideMPP": float,
           "re": re, "minidom": minidom, "_logger": logging.getLogger("golden"), "MPPExtractionError": MPPExtractionError}
    exec_defs(REF / "preprocessing" / "tiling.py", {"get_slide_mpp_", "_extract_mpp_from_comments", "_extract_mpp_from_metadata"}, glb)
    ome = '<OME><Image ID="Image:0"><Pixels PhysicalSizeX="0.4991" PhysicalSizeY="0.4991" SizeX="10"/></Image><Image><Pixels PhysicalSizeX="7.9"/></Image></OME>'
    cases = {"property": {"openslide.mpp-x": "0.2522", "openslide.comment": "<PixelSizeMicrons>0.9</PixelSizeMicrons>"},
             "comment": {"openslide.comment": "Aperio x|<PixelSizeMicrons>0.345</PixelSizeMicrons>|more"},
             "ome_xml": {"tiff.ImageDescription": ome},
             "comment_before_xml": {"openslide.comment": "<PixelSizeMicrons>0.5</PixelSizeMicrons>", "tiff.ImageDescription": ome},
             "bad_xml": {"tiff.ImageDescription": "not xml at all"},
             "xml_without_pixels": {"tiff.ImageDescription": "<OME><Image/></OME>"},
             "empty_comment": {"openslide.comment": ""},
             "nothing": {}}
    out = {}
    for name, props in cases.items():
        for default in (None, 0.75):
            rec = {"properties": props, "default_mpp": default}
            try:
                rec["mpp"] = float(glb["get_slide_mpp_"](types.SimpleNamespace(properties=props), default_mpp=default))
            except Exception as e:  # noqa: BLE001
                rec["error"] = type(e).__name__
            out[f"{name}|{default}"] = rec
    OUT.mkdir(parents=True, exist_ok=True)
    (OUT / "slide_mpp.json").write_text(json.dumps(out))
    print("wrote slide_mpp.json", {k: v.get("mpp", v.get("error")) for k, v in out.items()})


def main() -> None:
    install_shims()
    golden_chief()
    golden_eagle()
    golden_barspoon()
    golden_ticon()
    golden_keep_head()
    golden_plip()
    golden_dinov2_hf()
    golden_deploy_tables()
    golden_get_coords()
    golden_mil_vit()
    golden_mil_vit_train()
    golden_transmil()
    golden_mlp_cox_transforms()
    golden_bag()
    golden_bag_dataset()
    golden_ctranspath()
    golden_texture_gray()
    golden_tiling()
    golden_tile_cache()
    golden_slide_mpp()


if __name__ == "__main__":
    main()
