"""Encoder seam: gated-attention (CHIEF-style) slide / patient encoder on the HIP path.

Mirrors the reference's `Encoder` contract (src/stamp/encoding/encoder/__init__.py:29-229): constructor fields
``model, identifier, precision, required_extractors`` and the two methods the base class calls,
``_generate_slide_embedding(feats, device, coords=..., **kw) -> np.ndarray[D]`` (:164-171) and
``_generate_patient_embedding(feats_list, device, **kw)``.  The arithmetic is reference
src/stamp/encoding/encoder/chief.py:74-89, 255-275 (only ``WSI_feature = softmax(A) @ h_ori`` is consumed by STAMP,
chief.py:123-127).  Weights: the reference checkpoint's ``attention_net.*`` tensors.
"""
from __future__ import annotations

import hashlib
import logging
import os
import re
from pathlib import Path

import numpy as np
import torch

from . import h5io, ops

_logger = logging.getLogger("stamp_amd")
# `stamp_version` is parsed by the reference with packaging's `Version(...)` and compared with its own
# `stamp.__version__` (modeling/data.py:793-799): it must be a PEP 440 string not newer than the STAMP release whose file
# format this package writes (v2.5.0, pyproject.toml:3).  The build id of this package goes into `amdstamp_version`.
STAMP_FORMAT_VERSION = "2.5.0"
AMDSTAMP_VERSION = "0.3"
VERSION = STAMP_FORMAT_VERSION
# A trailing "-<at least six hex digits>" is the code hash STAMP appends to an extractor's name; everything in front of it is the name.
_HASHED_NAME = re.compile(r"(?P<base>.*)-[0-9a-fA-F]{6,}")


def resolve_extractor_name(name: str) -> str:
    """Behaviour of the reference's `_resolve_extractor_name` (encoder/__init__.py:235-250), pinned by tests/test_cpu_host.py against that
    function's own answers: surrounding blanks go, one trailing `-<hex hash>` goes, nothing else changes; the empty string is an error."""
    if not name:
        raise ValueError("Empty extractor name")
    stripped = str(name).strip()
    hashed = _HASHED_NAME.fullmatch(stripped)
    return hashed["base"] if hashed else stripped


def code_hash(directory: Path | None = None) -> str:
    """`get_processing_code_hash` semantics (reference utils/cache.py:42-55): sha256 over the sha256 digests of the sorted *.py files of a
    directory, here this package's (the reference's hash cannot be reproduced: different files); first 8 hex digits name output folders."""
    d = Path(directory) if directory else Path(__file__).resolve().parent
    h = hashlib.sha256()
    for f in sorted(d.glob("*.py")):
        h.update(hashlib.sha256(f.read_bytes()).digest())             # the raw 32-byte digests, like the reference (:51-54)
    return h.hexdigest()

_KEYS = {"fc_w": "attention_net.0.weight", "fc_b": "attention_net.0.bias",
         "a_w": "attention_net.3.attention_a.0.weight", "a_b": "attention_net.3.attention_a.0.bias",
         "b_w": "attention_net.3.attention_b.0.weight", "b_b": "attention_net.3.attention_b.0.bias",
         "c_w": "attention_net.3.attention_c.weight", "c_b": "attention_net.3.attention_c.bias"}


class HipGatedAttentionEncoder:
    def __init__(self, state_dict: dict[str, torch.Tensor], *, identifier: str = "chief",
                 required_extractors: tuple[str, ...] = ("chief-ctranspath",), device="cuda") -> None:
        self.identifier = identifier
        self.precision = torch.float32              # chief.py:117
        self.required_extractors = list(required_extractors)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipGatedAttentionEncoder runs on the GPU only (no CPU fallback)")
        missing = [v for v in _KEYS.values() if v not in state_dict]
        if missing:
            raise KeyError(f"state_dict lacks {missing}")
        self.weights = {k: state_dict[v].detach().to(self.device, torch.float32).contiguous() for k, v in _KEYS.items()}
        self.model = self                            # the base class only does .to(device).eval() on it

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def attention_raw(self, feats: torch.Tensor) -> torch.Tensor:
        return ops.gated_attn_pool(feats.to(self.device, torch.float32).contiguous(), self.weights, return_attn=True)[1]

    @torch.no_grad()
    def _generate_slide_embedding(self, feats: torch.Tensor, device=None, **kwargs) -> np.ndarray:
        if feats.dim() != 2 or feats.shape[0] == 0:
            raise ValueError(f"expected a non-empty [N, F] feature matrix, got {tuple(feats.shape)}")
        out = ops.gated_attn_pool(feats.to(self.device, torch.float32).contiguous(), self.weights)
        return out.detach().cpu().numpy()

    @torch.no_grad()
    def _generate_patient_embedding(self, feats_list: list, device=None, **kwargs) -> np.ndarray:
        return self._generate_slide_embedding(torch.cat([f.to(self.device) for f in feats_list], dim=0))

    @torch.no_grad()
    def _generate_slide_embeddings(self, feats_list: list) -> np.ndarray:
        """Several slides in ONE launch (amds_gated_attn_pool_batched): [len(feats_list), F].  Row i is what `_generate_slide_embedding(feats_list[i])`
        returns, up to the last-bit difference between the two decompositions of the fused kernel (include/amdstamp.h AMDS_GAP_AUTO)."""
        for f in feats_list:
            if f.dim() != 2 or f.shape[0] == 0:
                raise ValueError(f"expected a non-empty [N, F] feature matrix, got {tuple(f.shape)}")
        F, L, D = self.weights["fc_w"].shape[1], self.weights["fc_w"].shape[0], self.weights["a_w"].shape[0]
        if len(feats_list) == 1 or not ops.gated_attn_pool_batched_supported(F, L, D):
            return np.stack([self._generate_slide_embedding(f) for f in feats_list])
        x = torch.cat([f.to(self.device, torch.float32) for f in feats_list], dim=0).contiguous()
        return ops.gated_attn_pool_batched(x, [f.shape[0] for f in feats_list], self.weights).detach().cpu().numpy()

    # ---- the reference base class's file loops (encoder/__init__.py:42-229), on stamp_amd.h5io ---------------------------------------
    def _read_h5(self, h5_path: str):
        """-> (feats [N, F] in self.precision, CoordsInfo, extractor name without its hash suffix); what the reference's reader refuses
        (:182-201) is refused here with the same exception types: a path that is absent, a suffix other than .h5, a file without an
        `extractor` attribute."""
        path = Path(h5_path)
        if not path.exists():
            raise FileNotFoundError(f"File does not exist: {h5_path}")
        if path.suffix != ".h5":
            raise ValueError(f"File is not of type .h5: {path.name}")
        datasets, attrs = h5io.read_file(h5_path)
        named = attrs.get("extractor", "")
        if named == "":
            raise ValueError(f"Feature file does not have extractor's name in the metadata: {path.name}")
        feats = torch.from_numpy(np.ascontiguousarray(datasets["feats"])).to(dtype=self.precision)
        return feats, h5io.get_coords(datasets, attrs), resolve_extractor_name(named)

    def _require_extractor(self, got: str, accepted, h5_path, what: str = "Features") -> None:
        if got not in accepted:
            raise ValueError(f"{what} must be extracted with one of {list(accepted)}. Features located in {h5_path} are extracted with {got}")

    def _validate_and_read_features(self, h5_path: str):
        feats, coords, extractor = self._read_h5(h5_path)
        self._require_extractor(extractor, self.required_extractors, h5_path)
        return feats, coords

    def _save_features_(self, output_path: Path, feats: np.ndarray, feat_type: str) -> None:
        h5io.write_slide_features(output_path, feats, encoder=str(self.identifier), precision=str(self.precision), code_hash=code_hash()[:8],
                                  stamp_version=STAMP_FORMAT_VERSION, feat_type=feat_type, amdstamp_version=AMDSTAMP_VERSION)

    def encode_slides_(self, output_dir: Path, feat_dir: Path, device=None, generate_hash: bool = True, batch_slides: int = 32,
                       batch_rows: int = 1 << 19, **kwargs) -> None:
        """One slide-level .h5 per tile-level .h5 under feat_dir, folder structure kept, existing outputs skipped, files whose extractor
        is not accepted reported and skipped (:42-93).  The reference embeds one slide per iteration; here up to `batch_slides` slides (or
        `batch_rows` tiles) are read, pooled in ONE launch and written -- same files, same order.  batch_slides=1 is the per-slide loop."""
        encode_dir = Path(output_dir) / (f"{self.identifier}-slide-{code_hash()[:8]}" if generate_hash else f"{self.identifier}-slide")
        os.makedirs(encode_dir, exist_ok=True)
        feat_dir = Path(feat_dir)
        pending: list = []

        def flush() -> None:
            if pending:
                embs = self._generate_slide_embeddings([f for _, f in pending])
                for (path, _), emb in zip(pending, embs):
                    self._save_features_(path, emb, "slide")
                pending.clear()

        for h5_path in sorted(feat_dir.rglob("*.h5")):
            output_path = (encode_dir / h5_path.relative_to(feat_dir)).with_suffix(".h5")
            if output_path.exists():
                _logger.info(f"skipping {h5_path.stem} because {output_path} already exists")
                continue
            try:
                feats, coords = self._validate_and_read_features(str(h5_path))
            except ValueError as e:
                _logger.warning(str(e))
                continue
            if type(self)._generate_slide_embedding is not HipGatedAttentionEncoder._generate_slide_embedding:
                # a subclass with its own per-slide arithmetic (coords, second feature file): the per-slide loop
                self._save_features_(output_path, self._generate_slide_embedding(feats, device, coords=coords), "slide")
                continue
            if feats.dim() != 2 or feats.shape[0] == 0:
                flush()
                raise ValueError(f"expected a non-empty [N, F] feature matrix, got {tuple(feats.shape)}")
            pending.append((output_path, feats))
            if len(pending) >= max(1, batch_slides) or sum(f.shape[0] for _, f in pending) >= batch_rows:
                flush()
        flush()

    def encode_patients_(self, output_dir: Path, feat_dir: Path, patient_to_files: dict[str, list[str]], device=None, generate_hash: bool = True,
                         **kwargs) -> None:
        """One patient-level .h5 per patient from all of the patient's tile-feature files (:95-162).  `patient_to_files` is the grouping the
        reference derives from its slide table (`read_table(...).groupby(patient_label)[filename_label]`): table I/O stays with the caller."""
        encode_dir = Path(output_dir) / (f"{self.identifier}-pat-{code_hash()[:8]}" if generate_hash else f"{self.identifier}-pat")
        os.makedirs(encode_dir, exist_ok=True)
        for patient_id, files in patient_to_files.items():
            output_path = (encode_dir / str(patient_id)).with_suffix(".h5")
            if output_path.exists():
                _logger.info(f"skipping {patient_id} because {output_path} already exists")
                continue
            feats_list = [self._validate_and_read_features(os.path.join(feat_dir, f))[0] for f in files]
            if not feats_list:
                _logger.warning(f"No features found for patient {patient_id}, skipping.")
                continue
            self._save_features_(output_path, self._generate_patient_embedding(feats_list, device, **kwargs), "patient")


def align_by_coords(ref_coords_um: np.ndarray, other_coords_um: np.ndarray, decimals: int = 5) -> np.ndarray:
    """Permutation `perm` with other[perm[i]] at ref[i]'s coordinate after rounding to `decimals` -- the contract of the reference's
    `_align_vir2_to_ctp_by_coords` (eagle.py:265-300; pinned by tests/golden/eagle.npz): equal coordinates pair up in file order, a coordinate
    of `ref` that `other` lacks (or has fewer times) raises ValueError, and so does any row of `other` left unmatched.

    Vectorised: both sets are labelled by their row in the sorted union of distinct coordinates, then each side is ranked by a STABLE sort
    on (label, file position); the k-th occurrence of a label in `ref` receives the k-th occurrence of it in `other`."""
    ref = np.round(np.asarray(ref_coords_um, dtype=np.float64), decimals).reshape(len(ref_coords_um), -1)
    oth = np.round(np.asarray(other_coords_um, dtype=np.float64), decimals).reshape(len(other_coords_um), -1)
    n_ref, n_oth = ref.shape[0], oth.shape[0]
    both = np.concatenate([ref, oth], axis=0) + 0.0                                  # + 0.0: -0.0 and 0.0 are one coordinate
    _, label = np.unique(both, axis=0, return_inverse=True)
    label = np.asarray(label).reshape(-1)
    n_labels = int(label.max()) + 1 if label.size else 0
    have = np.bincount(label[n_ref:], minlength=n_labels)
    want = np.bincount(label[:n_ref], minlength=n_labels)
    short = np.flatnonzero(want > have)
    if short.size:
        # report the first row of `ref` (file order) that cannot be served
        seen = np.zeros(n_labels, dtype=np.int64)
        for i in range(n_ref):
            seen[label[i]] += 1
            if seen[label[i]] > have[label[i]]:
                raise ValueError(f"Missing coord in other set: {tuple(ref[i].tolist())}")
    extra = int((have - want).sum())
    if extra:
        raise ValueError(f"virchow2 features contain {extra} extra coords not in ref.")
    ref_order = np.argsort(label[:n_ref], kind="stable")
    oth_order = np.argsort(label[n_ref:], kind="stable")
    perm = np.empty(n_ref, dtype=np.int64)
    perm[ref_order] = oth_order
    return perm


class HipEagleEncoder(HipGatedAttentionEncoder):
    """The reference's EAGLE slide / patient encoder (src/stamp/encoding/encoder/eagle.py:28-263) on the HIP path: CHIEF's gated-attention scores
    of a slide's CTransPath tile features (`amds_gated_attn_pool`'s attention_raw) pick the 25 highest-scoring tiles, the embedding is the mean
    of THOSE tiles' Virchow2 features (`amds_topk_rows_mean`: selection and mean in one launch).  Same seam as the reference class:
    `_generate_slide_embedding(feats, device, agg_feats=)`, `_generate_patient_embedding(feats_list, device, agg_feats_list=)`,
    `encode_slides_(..., agg_feat_dir=)`, `encode_patients_(..., agg_feat_dir=)`; the two feature files of a slide must describe the same
    tiles -- a permuted second file is re-ordered by coordinates (:66-81), anything else is an error for that slide."""
    TOP_K = 25                                                                                   # eagle.py:107

    def __init__(self, state_dict: dict[str, torch.Tensor], *, identifier: str = "eagle", required_extractors: tuple[str, ...] = ("ctranspath", "chief-ctranspath"),
                 required_agg_extractor: str = "virchow2", device="cuda") -> None:
        super().__init__(state_dict, identifier=identifier, required_extractors=required_extractors, device=device)
        self.required_agg_extractor = required_agg_extractor                                      # eagle.py:30

    def top_tiles(self, feats: torch.Tensor) -> torch.Tensor:
        """Indices (descending score) of the tiles EAGLE keeps."""
        araw = self.attention_raw(feats)
        k = min(self.TOP_K, araw.shape[0])
        return ops.topk_rows_mean(araw, feats.to(self.device, torch.float32).contiguous(), k)[0].long()

    @torch.no_grad()
    def _generate_slide_embedding(self, feats: torch.Tensor, device=None, agg_feats: torch.Tensor | None = None, **kwargs) -> np.ndarray:
        if agg_feats is None:
            raise ValueError("agg_feats is required for slide embedding")                        # eagle.py:98-99
        if feats.dim() != 2 or feats.shape[0] == 0:
            raise ValueError(f"expected a non-empty [N, F] feature matrix, got {tuple(feats.shape)}")
        if agg_feats.dim() != 2 or agg_feats.shape[0] != feats.shape[0]:
            raise ValueError(f"agg_feats must have one row per tile: {tuple(agg_feats.shape)} vs {tuple(feats.shape)}")
        araw = self.attention_raw(feats)
        agg = agg_feats.to(self.device)
        if agg.dtype not in (torch.float32, torch.float16):
            agg = agg.float()
        k = min(self.TOP_K, araw.shape[0])
        _, mean = ops.topk_rows_mean(araw, agg.contiguous(), k)
        return mean.detach().cpu().numpy()

    @torch.no_grad()
    def _generate_patient_embedding(self, feats_list: list, device=None, agg_feats_list: list | None = None, **kwargs) -> np.ndarray:
        if agg_feats_list is None:
            raise ValueError("agg_feats_list is required for patient embedding")                 # eagle.py:129-130
        return self._generate_slide_embedding(torch.cat([f.to(self.device) for f in feats_list], dim=0),
                                              agg_feats=torch.cat([f.to(self.device) for f in agg_feats_list], dim=0))

    def _validate_and_read_features_with_agg(self, h5_ctp: str, h5_vir2: str, slide_name: str):
        """(:41-94) both files read and validated; the second re-ordered to the first's tile order when it is a permutation of it."""
        feats, coords, extractor = self._read_h5(h5_ctp)
        if extractor not in self.required_extractors:
            raise ValueError(f"Features must be extracted with one of {self.required_extractors}. Features located in {h5_ctp} are extracted with {extractor}")
        agg_feats, agg_coords, extractor = self._read_h5(h5_vir2)
        if extractor != self.required_agg_extractor:
            raise ValueError(f"Aggregated features must be extracted with {self.required_agg_extractor} Features located in {h5_vir2} are extracted with {extractor}")
        c0, c1 = np.asarray(coords.coords_um), np.asarray(agg_coords.coords_um)
        if c0.shape != c1.shape or not np.allclose(c0, c1, atol=1e-5, rtol=0):
            try:
                perm = align_by_coords(c0, c1, decimals=5)
            except ValueError as e:
                raise ValueError(f"Coordinates mismatch between ctranspath and virchow2 features for slide {slide_name}. Alignment attempt failed: {e}")
            agg_feats, c1 = agg_feats[torch.from_numpy(perm)], c1[perm]
            if not np.allclose(c0, c1, atol=1e-5, rtol=0):
                raise ValueError(f"Coordinates mismatch between ctranspath and virchow2 features for slide {slide_name}. Ensure that both are aligned.")
        return feats, agg_feats

    def encode_slides_(self, output_dir: Path, feat_dir: Path, device=None, generate_hash: bool = True, **kwargs) -> None:
        """(:136-185) one slide-level file per tile-feature file of `feat_dir` whose twin exists in `agg_feat_dir`; existing outputs skipped,
        slides whose two files do not fit together reported and skipped."""
        agg_feat_dir = kwargs.get("agg_feat_dir")
        if not agg_feat_dir:
            raise ValueError("agg_feat_dir that contains virchow2 features is required for Eagle's encode_slides")
        encode_dir = Path(output_dir) / (f"{self.identifier}-slide-{code_hash()[:8]}" if generate_hash else f"{self.identifier}-slide")
        os.makedirs(encode_dir, exist_ok=True)
        for name in sorted(os.listdir(feat_dir)):
            output_path = (encode_dir / Path(name).name).with_suffix(".h5")
            if output_path.exists():
                _logger.info(f"skipping {name} because {output_path} already exists")
                continue
            try:
                feats, agg = self._validate_and_read_features_with_agg(os.path.join(feat_dir, name), os.path.join(agg_feat_dir, name), Path(name).name)
            except ValueError as e:
                _logger.warning(str(e))
                continue
            self._save_features_(output_path, self._generate_slide_embedding(feats, device, agg_feats=agg), "slide")

    def encode_patients_(self, output_dir: Path, feat_dir: Path, patient_to_files: dict[str, list[str]], device=None, generate_hash: bool = True,
                         **kwargs) -> None:
        """(:188-262) one patient-level file from all of the patient's slides; a slide whose file is missing is skipped, a patient with no slide
        left is reported and skipped.  `patient_to_files`: the grouping of the reference's slide table."""
        agg_feat_dir = kwargs.get("agg_feat_dir")
        if not agg_feat_dir:
            raise ValueError("agg_feat_dir that contains virchow2 features is required for Eagle's encode_patients")
        encode_dir = Path(output_dir) / (f"{self.identifier}-pat-{code_hash()[:8]}" if generate_hash else f"{self.identifier}-pat")
        os.makedirs(encode_dir, exist_ok=True)
        for patient_id, files in patient_to_files.items():
            output_path = (encode_dir / str(patient_id)).with_suffix(".h5")
            if output_path.exists():
                _logger.info(f"skipping {patient_id} because {output_path} already exists")
                continue
            feats_list, agg_list = [], []
            for f in files:
                try:
                    feats, agg = self._validate_and_read_features_with_agg(os.path.join(feat_dir, f), os.path.join(agg_feat_dir, f), Path(f).stem)
                except FileNotFoundError as e:
                    _logger.warning(f"[{patient_id}] skip slide (FileNotFoundError): {Path(f).stem} -> {e}")
                    continue
                feats_list.append(feats)
                agg_list.append(agg)
            if not feats_list:
                _logger.warning(f"No ctranspath features for patient {patient_id}")
                continue
            self._save_features_(output_path, self._generate_patient_embedding(feats_list, device, agg_feats_list=agg_list), "patient")


class HipTitanShapedEncoder(HipGatedAttentionEncoder):
    """A STAND-IN with TITAN's interface and tensor shapes -- NOT TITAN's arithmetic.  **Parity unpinned by construction.**

    The reference's `Titan` encoder (src/stamp/encoding/encoder/titan.py:28-61) calls `model.encode_slide_from_patch_features(feats,
    coords_px, patch_size_lvl0)` of `AutoModel.from_pretrained("MahmoodLab/TITAN", trust_remote_code=True)`: the model's code and weights
    live on the Hugging Face hub, not in the repository, and cannot be fetched here -- there is nothing to restate and nothing to compare
    against.  What IS in the repository is the seam: 768-d CONCH1.5 tile features + integer level-0 pixel coordinates in, one 768-d slide
    vector out (float32, `required_extractors = [conch1_5]`), the per-patient "virtual slide" built by concatenating a patient's slides
    along x with a running offset (:87-179).  This class implements that seam on the HIP path with a transformer of TITAN's published
    geometry (6 layers, 12 heads of 64, width 768, feed-forward 3072, attention biased by 2-D tile distance) assembled from this package's
    own MIL `vit` blocks with their post-softmax ALiBi -- so that BASELINE.json configs[4]'s slide-encoding stage has a same-shape
    workload to measure, and so that a maintainer with hub access has the slot real TITAN weights + its own attention rule would fill.
    Features it writes carry `encoder = "titan-standin"`, never "titan"."""

    GEOMETRY = dict(dim_input=768, dim_model=768, n_layers=6, n_heads=12, dim_feedforward=3072, dim_output=768)

    def __init__(self, state_dict: dict[str, torch.Tensor] | None = None, *, seed: int = 0, identifier: str = "titan-standin",
                 required_extractors: tuple[str, ...] = ("conch1_5",), device="cuda") -> None:
        from .mil import VisionTransformer
        self.identifier = identifier
        self.precision = torch.float32              # titan.py:33
        self.required_extractors = list(required_extractors)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipTitanShapedEncoder runs on the GPU only (no CPU fallback)")
        g = torch.Generator().manual_seed(seed)
        rng_state = torch.get_rng_state()
        torch.manual_seed(int(torch.randint(0, 2 ** 31, (1,), generator=g)))
        self.net = VisionTransformer(dropout=0.0, use_alibi=True, **self.GEOMETRY).eval()
        torch.set_rng_state(rng_state)
        if state_dict is not None:
            self.net.load_state_dict(state_dict)
        self.net = self.net.to(self.device)
        self.model = self

    def attention_raw(self, feats):
        raise NotImplementedError("the TITAN-shaped stand-in has no gated-attention scores")

    @torch.no_grad()
    def _generate_slide_embedding(self, feats: torch.Tensor, device=None, coords=None, **kwargs) -> np.ndarray:
        if coords is None:
            raise ValueError("Coords must be provided.")                                  # titan.py:46-47
        if feats.dim() == 3 and feats.shape[0] == 1:                                      # the patient path hands [1, N, F] (titan.py:73)
            feats = feats[0]
        if feats.dim() != 2 or feats.shape[0] == 0 or feats.shape[1] != self.GEOMETRY["dim_input"]:
            raise ValueError(f"expected a non-empty [N, 768] feature matrix, got {tuple(feats.shape)}")
        # titan.py:49-53: micrometres -> level-0 pixels, truncated to int64; here additionally expressed in tiles (the distance unit of the bias)
        coords_px = (torch.tensor(np.asarray(coords.coords_um), dtype=self.precision) / coords.mpp).to(torch.int64)
        grid = coords_px.to(torch.float32) / float(int(coords.tile_size_px))
        out = self.net(feats.to(self.device, torch.float32)[None].contiguous(), coords=grid.to(self.device)[None].contiguous(), mask=None)
        return out.detach().squeeze().cpu().numpy()

    @torch.no_grad()
    def _generate_patient_embedding(self, feats_list: list, device=None, coords_list=None, **kwargs) -> np.ndarray:
        if coords_list is None:
            raise ValueError("coords_list must be provided.")                             # titan.py:70-71
        cat = torch.cat([f.to(self.device) for f in feats_list], dim=0).unsqueeze(0)
        coords = h5io.CoordsInfo(np.concatenate([c.coords_um for c in coords_list], axis=0), coords_list[0].tile_size_um, coords_list[0].tile_size_px)
        return self._generate_slide_embedding(cat, device, coords)

    def encode_patients_(self, output_dir: Path, feat_dir: Path, patient_to_files: dict[str, list[str]], device=None, generate_hash: bool = True,
                         **kwargs) -> None:
        """titan.py:87-179: one virtual slide per patient, the patient's slides laid side by side along x (offset = rightmost tile's x + tile
        width of everything placed so far); all slides of a patient must share one mpp."""
        import math
        encode_dir = Path(output_dir) / (f"{self.identifier}-pat-{code_hash()[:8]}" if generate_hash else f"{self.identifier}-pat")
        os.makedirs(encode_dir, exist_ok=True)
        for patient_id, files in patient_to_files.items():
            output_path = (encode_dir / str(patient_id)).with_suffix(".h5")
            if output_path.exists():
                _logger.info(f"skipping {patient_id} because {output_path} already exists")
                continue
            feats_list, coords_list, x_off, mpp = [], [], 0.0, -1.0
            for name in files:
                if not str(name).endswith(".h5"):
                    _logger.warning(f"Skipping {name} (not an .h5 file)")
                    continue
                try:
                    feats, coords = self._validate_and_read_features(os.path.join(feat_dir, name))
                except (FileNotFoundError, ValueError, OSError) as e:
                    _logger.warning(f"Skipping {name}: {e}")
                    continue
                if mpp < 0:
                    mpp = coords.mpp
                elif not math.isclose(mpp, coords.mpp, rel_tol=1e-5):
                    raise ValueError("All patient slides must have the same mpp value. Try reprocessing the slides using the same tile_size_um and "
                                     "tile_size_px values for all of them.")
                cu = np.array(coords.coords_um, dtype=np.float64, copy=True)
                cu[:, 0] += x_off
                x_off = float(cu[:, 0].max()) + float(coords.tile_size_um)
                feats_list.append(feats)
                coords_list.append(h5io.CoordsInfo(cu, coords.tile_size_um, coords.tile_size_px))
            if not feats_list:
                _logger.warning(f"No features found for patient {patient_id}, skipping.")
                continue
            self._save_features_(output_path, self._generate_patient_embedding(feats_list, device, coords_list), "patient")
