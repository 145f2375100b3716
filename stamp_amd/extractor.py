"""Extractor seam: the object `stamp preprocess` accepts via ``extract_(extractor=<Extractor>)``
(reference src/stamp/preprocessing/__init__.py:118, 237-238).

`Extractor` mirrors the reference dataclass field for field (src/stamp/preprocessing/extractor/__init__.py:17-28:
keyword-only ``model``, ``transform``, ``identifier``) so an instance built here can be handed to an unmodified
STAMP; when STAMP itself is importable, build `stamp.preprocessing.extractor.Extractor` with the same three values.

`transform` does NOT normalise on the host: it returns the decoded tile as a ``uint8 [H, W, 3]`` tensor; the
``(x/255-mean)/std`` step of the reference's transforms (h_optimus_0.py:22-30) is folded into the HIP patch
embedding.  The reference's loop only does ``model(tiles.to(device)).detach().half().cpu()`` on whatever the
transform produced (:324-325), so the dtype change is invisible to it.
"""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass

import numpy as np
import torch

from .swin import SWIN_PRESETS, HipSwin, SwinConfig
from .vit import PRESETS, HipViT, ViTConfig


@dataclass(frozen=True, kw_only=True)
class Extractor:
    model: torch.nn.Module
    transform: Callable[[object], torch.Tensor]
    identifier: str


def u8_tile_transform(img) -> torch.Tensor:
    """PIL.Image / ndarray tile -> uint8 [H, W, 3] tensor (no float conversion, no normalisation)."""
    arr = np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)
    if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError(f"expected an RGB uint8 tile, got {arr.dtype} {arr.shape}")
    return torch.from_numpy(np.ascontiguousarray(arr))


def hip_vit_extractor(name: str, state_dict: dict[str, torch.Tensor], *, identifier: str | None = None,
                      cfg: ViTConfig | None = None, device="cuda", act_dtype=torch.float16, chunk: int = 1020) -> Extractor:
    """Extractor whose model is the HIP tile encoder.  `name` is a key of `stamp_amd.vit.PRESETS`; `state_dict`
    uses timm VisionTransformer names (what the reference's factories load, e.g. uni2.py:32-34)."""
    cfg = cfg or PRESETS[name]
    model = HipViT(cfg, state_dict, device=device, act_dtype=act_dtype, chunk=chunk)
    return Extractor(model=model, transform=u8_tile_transform, identifier=identifier or f"amdstamp-{name}")


def hip_ctranspath_extractor(state_dict: dict[str, torch.Tensor], *, identifier: str = "ctranspath", cfg: SwinConfig | None = None,
                             device="cuda", act_dtype=torch.float16, chunk: int = 1024) -> Extractor:
    """The reference's `ctranspath()` / `chief_ctranspath()` factories (src/stamp/preprocessing/extractor/ctranspath.py:34-70,
    chief_ctranspath.py:20-57) with the HIP model: `state_dict` is what they pass to `model.load_state_dict`
    (`torch.load("ctranspath.pth")["model"]`, after the sha256 check they do), `identifier` the ExtractorName value
    ("ctranspath" or "chief-ctranspath") so that the CHIEF slide encoder accepts the feature files."""
    model = HipSwin(cfg or SWIN_PRESETS["ctranspath"], state_dict, device=device, act_dtype=act_dtype, chunk=chunk)
    return Extractor(model=model, transform=u8_tile_transform, identifier=identifier)


def has_enough_texture(tiles_u8: torch.Tensor, cutoff: float = 0.02) -> torch.Tensor:
    """Batched `_has_enough_texture` (reference src/stamp/preprocessing/tiling.py:280-291) for decoded tiles already on
    the GPU: bool [B], True = keep (Canny edge fraction >= canny_cutoff; 40 / 100 are the reference's hard-coded thresholds)."""
    from . import ops
    return ops.tile_edge_fraction(tiles_u8, 40, 100) >= cutoff


@torch.inference_mode()
def extract_tiles(extractor: Extractor, tiles_u8: torch.Tensor, batch_size: int = 1020, device="cuda") -> torch.Tensor:
    """The reference's per-slide hot loop (preprocessing/__init__.py:322-327) on an in-memory stack of decoded tiles:
    batches -> model -> ``.half()`` -> host.  The reference's batch of 64 is a host-RAM choice; the HIP model
    chunks internally, so larger batches only amortise the PCIe copy."""
    model = extractor.model
    outs = []
    for i in range(0, tiles_u8.shape[0], batch_size):
        outs.append(model(tiles_u8[i:i + batch_size].to(device, non_blocking=True)).detach().half().cpu())
    if not outs:
        return torch.empty(0, getattr(model.cfg, "out_dim", None) or model.cfg.dim, dtype=torch.float16)
    return torch.cat(outs)
