"""One slide through the whole extraction path on the GPU: supertiles -> tiles -> texture filter -> tile encoder -> feature .h5.

Mirrors the per-slide body of the reference's `extract_` (src/stamp/preprocessing/__init__.py:275-367) with the tile producer of
`tiling.py` (`_supertiles` :294-347, `_tiles` :196-247, `_tiles_with_tissue` :171-193) in front of it:

    slide.get_thumbnail -> foreground supertiles (host, a few hundred pixels)                         tiling.foreground_coords
    slide.read_region RGBA [S, S, 4], `max_workers` reader threads (host I/O, as the reference)      ThreadPoolExecutor
      -> pinned batch -> H2D -> PIL-exact bicubic resize + crop into 224 x 224 x 3 tiles (HIP)       tiling.supertiles_to_tiles
      -> Canny edge fraction >= canny_cutoff (HIP)                                                   extractor.has_enough_texture
      -> tile encoder (HIP) -> fp16 features                                                         Extractor.model
    feats fp16 [N, D] + coords f32 [N, 2] (um) + attributes -> .h5 in STAMP's schema                h5io.write_tile_features

`slide` is anything with openslide's `dimensions`, `read_region((x, y), 0, (w, h)) -> RGBA PIL image` and `get_thumbnail(size)`;
opening slide files is the caller's business (openslide / the reference's own reader).  Tiles come out in supertile order (the
reference yields in thread-completion order, :346, which is why its tests sort by coordinate before comparing).
"""
from __future__ import annotations

from concurrent import futures
from pathlib import Path

import numpy as np
import torch

from . import h5io, tiling
from .encoder import AMDSTAMP_VERSION, STAMP_FORMAT_VERSION, code_hash
from .extractor import Extractor, has_enough_texture


def _region_array(slide, x: int, y: int, s: int) -> np.ndarray:
    return np.asarray(slide.read_region((x, y), 0, (s, s)).convert("RGBA"), dtype=np.uint8)


@torch.inference_mode()
def extract_slide(slide, extractor: Extractor, output_path, *, slide_mpp: float, tile_size_um: float = 256.0, tile_size_px: int = 224,
                  max_supertile_size_slide_px: int = 2 ** 10, brightness_cutoff: int | None = 240, canny_cutoff: float | None = 0.02,
                  max_workers: int = 8, supertiles_per_batch: int = 16, device="cuda") -> dict:
    """Writes `output_path` (nothing if the slide has no tiles, like the reference :338-340) and returns counters.
    Defaults are the reference's (preprocessing/config.py:46-66; max_supertile_size_slide_px = 2**10 at __init__.py:307)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("extract_slide runs on the GPU only (no CPU fallback)")
    geo = tiling.supertile_geometry(slide_mpp, tile_size_um, tile_size_px, max_supertile_size_slide_px)
    S, k = geo.supertile_size_slide_px, geo.tiles_per_side
    dims = tuple(int(v) for v in slide.dimensions)
    gw, gh = tiling.thumbnail_size(dims, S)
    origins = tiling.foreground_coords(dims, slide.get_thumbnail((2 * gw, 2 * gh)), S, brightness_cutoff)
    feats, coords = [], []
    n_tiles = 0
    model = extractor.model
    host = torch.empty(supertiles_per_batch, S, S, 4, dtype=torch.uint8).pin_memory()
    with futures.ThreadPoolExecutor(max_workers) as pool:
        for i in range(0, len(origins), supertiles_per_batch):
            batch = origins[i:i + supertiles_per_batch]
            for j, arr in enumerate(pool.map(lambda o: _region_array(slide, o[0], o[1], S), batch)):
                host[j].copy_(torch.from_numpy(arr))
            tiles = tiling.supertiles_to_tiles(host[:len(batch)].to(dev, non_blocking=True), k, tile_size_px)
            cu = np.concatenate([tiling.tile_coords_um(o, slide_mpp, k, tile_size_um) for o in batch])
            n_tiles += tiles.shape[0]
            if canny_cutoff is not None:
                keep = has_enough_texture(tiles, canny_cutoff)
                tiles, cu = tiles[keep], cu[keep.cpu().numpy()]
            if tiles.shape[0]:
                feats.append(model(tiles.contiguous()).detach().half().cpu())
                coords.append(cu)
    stats = {"supertiles": len(origins), "tiles_seen": n_tiles, "tiles_kept": int(sum(f.shape[0] for f in feats))}
    if not feats:
        return stats
    h5io.write_tile_features(Path(output_path), torch.cat(feats), np.concatenate(coords).astype(np.float32), extractor=str(extractor.identifier),
                             tile_size_um=tile_size_um, tile_size_px=tile_size_px, code_hash=code_hash()[:8], stamp_version=STAMP_FORMAT_VERSION,
                             amdstamp_version=AMDSTAMP_VERSION)
    return stats
