"""Tiling of a whole-slide image on the step before the tile encoder: supertile geometry, background rejection on the thumbnail,
supertile -> tiles on the GPU, tile coordinates.  Mirrors reference src/stamp/preprocessing/tiling.py (`_supertiles` :294-347,
`_foreground_coords` :250-277, `_tiles` :196-247).  Reading pixels out of a slide file (openslide) and the thumbnail are host I/O
and stay with the caller; what happens to the pixels afterwards runs here:

    read_region RGBA [S, S, 4] u8  --amds_supertiles_to_tiles_u8-->  k x k tiles [224, 224, 3] u8  (+ micrometre coordinates)

The GPU resize is PIL's `Image.resize` bit for bit (bicubic on the premultiplied image, 8-bit two-pass resample with 22-bit
fixed-point taps; see csrc/resize.hip), so tiles equal the reference's `supertile.resize(...).convert("RGB").crop(...)`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from functools import lru_cache

import numpy as np
import torch

from . import _lib, ops


class MPPExtractionError(Exception):
    """No microns-per-pixel value anywhere in the slide's metadata and no default given (the reference's exception of the same name, tiling.py:46-50)."""


def _mpp_from_property(props) -> float | None:
    v = props.get("openslide.mpp-x")                       # openslide.PROPERTY_NAME_MPP_X
    return None if v is None else float(v)


def _mpp_from_comment(props) -> float | None:
    import re
    hit = re.search(r"<PixelSizeMicrons>(.*?)</PixelSizeMicrons>", props.get("openslide.comment", "") or "")
    return float(hit.group(1)) if hit else None


def _mpp_from_image_description(props) -> float | None:
    """OME-style XML in the TIFF ImageDescription: PhysicalSizeX of the first <Pixels> of the first <Image>; anything malformed counts as 'not there'."""
    text = props.get("tiff.ImageDescription") or None
    if text is None:
        return None
    try:
        from xml.dom import minidom
        root = minidom.parseString(text).documentElement
        return float(root.getElementsByTagName("Image")[0].getElementsByTagName("Pixels")[0].getAttribute("PhysicalSizeX"))
    except Exception:
        return None


def get_slide_mpp(slide, *, default_mpp: float | None = None) -> float:
    """Microns per pixel of a slide, looked up where the reference's `get_slide_mpp_` looks, in its order (tiling.py:409-446): the `openslide.mpp-x` property;
    `<PixelSizeMicrons>` in `openslide.comment`; `PhysicalSizeX` in the OME XML of `tiff.ImageDescription`; then `default_mpp` (with a warning); else
    MPPExtractionError.  `slide`: an object with a `properties` mapping (an opened openslide slide), such a mapping itself, or a path (opened with
    openslide, which this package does not depend on otherwise).  As in the reference, a source that yields 0 counts as absent for the later sources."""
    import logging
    import os
    opened = None
    if isinstance(slide, (str, os.PathLike)):
        import openslide                                     # reference behaviour for a path; ImportError says what is missing
        slide = opened = openslide.open_slide(slide)
    try:
        props = slide if hasattr(slide, "get") and not hasattr(slide, "properties") else slide.properties
        if props.get("openslide.mpp-x") is not None:
            mpp = _mpp_from_property(props)                  # (the reference takes this value as it is, even 0)
        else:
            mpp = _mpp_from_comment(props) or _mpp_from_image_description(props)
    finally:
        if opened is not None and hasattr(opened, "close"):
            opened.close()
    if mpp:
        return float(mpp)
    if default_mpp:
        logging.getLogger("stamp").warning(f"could not infer slide MPP from metadata, using {default_mpp} instead.")
        return float(default_mpp)
    raise MPPExtractionError()


@dataclass(frozen=True)
class SupertileGeometry:
    tile_size_slide_px: int          # one tile's side in level-0 pixels, ceil(tile_size_um / mpp)            (tiling.py:311)
    tiles_per_side: int              # k = max(int(max_supertile_px * mpp // tile_size_um), 1)                 (:308-309)
    supertile_size_slide_px: int     # the square read from the slide                                          (:312-314)
    supertile_size_tile_px: int      # its side after the resize, k * tile_size_px                            (:315)
    supertile_size_um: float         # (:317)


def supertile_geometry(slide_mpp: float, tile_size_um: float = 256.0, tile_size_px: int = 224, max_supertile_size_slide_px: int = 4096) -> SupertileGeometry:
    max_supertile_um = max_supertile_size_slide_px * slide_mpp
    k = max(int(max_supertile_um // tile_size_um), 1)
    tile_px_slide = int(np.ceil(tile_size_um / slide_mpp))
    s = tile_px_slide * k
    return SupertileGeometry(tile_px_slide, k, s, tile_size_px * k, s * slide_mpp)


def thumbnail_size(dimensions: tuple[int, int], supertile_size_slide_px: int) -> tuple[int, int]:
    """(width, height) of the one-cell-per-supertile grid; the reference asks the slide for a thumbnail of twice that (:254-261)."""
    g = np.ceil(np.array(dimensions) / supertile_size_slide_px).astype(np.uint32)
    return int(g[0]), int(g[1])


def foreground_coords(dimensions: tuple[int, int], thumbnail_2x, supertile_size_slide_px: int, brightness_cutoff: int | None) -> list[tuple[int, int]]:
    """Level-0 origins (x, y) of the supertiles that are not background, row-major -- `_foreground_coords` (:250-277).
    `thumbnail_2x`: what `slide.get_thumbnail(2 * grid)` returned (a PIL image); like the reference it is resized to the grid and
    converted to 32-bit luma with PIL (host glue on a few hundred pixels)."""
    gw, gh = thumbnail_size(dimensions, supertile_size_slide_px)
    gray = np.array(thumbnail_2x.resize((gw, gh)).convert("I"))
    fg = gray < brightness_cutoff if brightness_cutoff is not None else np.ones_like(gray, dtype=bool)
    return [(x, y) for y in range(0, dimensions[1], supertile_size_slide_px) for x in range(0, dimensions[0], supertile_size_slide_px)
            if fg[y // supertile_size_slide_px, x // supertile_size_slide_px]]


def tile_coords_um(origin_slide_px: tuple[int, int], slide_mpp: float, tiles_per_side: int, tile_size_um: float) -> np.ndarray:
    """float64 [k*k, 2] top-left micrometre coordinates of a supertile's tiles in the reference's yield order (y outer, x inner; :237-246)."""
    ox, oy = origin_slide_px[0] * slide_mpp, origin_slide_px[1] * slide_mpp
    return np.array([(ox + x * tile_size_um, oy + y * tile_size_um) for y in range(tiles_per_side) for x in range(tiles_per_side)], dtype=np.float64)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=16)
def resize_coefficients(in_size: int, out_size: int) -> tuple[np.ndarray, np.ndarray]:
    """Pillow's tap table for a bicubic resize in_size -> out_size (Resample.c precompute_coeffs / normalize_coeffs_8bpc, evaluated in
    double precision in the same operation order): bounds int32 [out, 2] = (first tap, tap count), taps int32 [out, ksize], 22
    fraction bits.  Host-side set-up, once per (S, S') pair."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    taps = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            taps[xx, x] = int(v * (1 << 22) - 0.5) if v < 0 else int(v * (1 << 22) + 0.5)
        bounds[xx] = (xmin, n)
    return bounds, taps


@lru_cache(maxsize=16)
def _resize_tables_on_device(in_size: int, out_size: int, device_index: int):
    """The tap tables of `resize_coefficients` as device tensors, uploaded once per (S, S', device): per call they would be two synchronous
    pageable H2D copies in front of every batch of supertiles."""
    bounds, taps = resize_coefficients(in_size, out_size)
    dev = torch.device("cuda", device_index)
    return torch.from_numpy(bounds).to(dev), torch.from_numpy(taps).to(dev), int(taps.shape[1])


def resize_center_crop(tiles: torch.Tensor, resized: int, crop: int) -> torch.Tensor:
    """torchvision's `Resize(resized, BICUBIC)` + `CenterCrop(crop)` on square RGB u8 tiles [n, S, S, 3] on the GPU -> u8 [n, crop, crop, 3], bit for bit
    what the reference's transform produces from the PIL tile (gigapath.py:21-28): Pillow's bicubic resample + torchvision's crop offset."""
    if not tiles.is_cuda:
        raise RuntimeError("resize_center_crop needs the tiles on the GPU (no CPU fallback)")
    if tiles.dtype != torch.uint8 or tiles.dim() != 4 or tiles.shape[-1] != 3 or tiles.shape[1] != tiles.shape[2]:
        raise ValueError(f"expected u8 [n, S, S, 3], got {tiles.dtype} {tuple(tiles.shape)}")
    if crop > resized:
        raise ValueError(f"crop {crop} larger than the resized image {resized}")
    tiles = tiles.contiguous()
    n, S = tiles.shape[0], tiles.shape[1]
    dev = tiles.device
    b_d, t_d, ksize = _resize_tables_on_device(S, int(resized), dev.index if dev.index is not None else torch.cuda.current_device())
    lib = _lib.lib()
    out = torch.empty(n, crop, crop, 3, dtype=torch.uint8, device=dev)
    for i0 in range(0, n, 32768):                          # gridDim.z of one launch
        m = min(32768, n - i0)
        nb = lib.amds_tile_resize_crop_workspace_bytes(m, S, crop)
        ws = ops.scratch("resize_crop", dev, nb)
        _lib.check(lib.amds_tile_resize_crop_u8(tiles[i0:i0 + m].data_ptr(), out[i0:i0 + m].data_ptr(), m, S, int(resized), int(crop), b_d.data_ptr(), t_d.data_ptr(), ksize,
                                                ws.data_ptr(), ws.numel(), ops._stream()), "tile_resize_crop")
    return out


def supertiles_to_tiles(rgba: torch.Tensor, tiles_per_side: int, tile_size_px: int = 224, *, out: torch.Tensor | None = None,
                        workspace: torch.Tensor | None = None) -> torch.Tensor:
    """u8 [n, S, S, 4] (what `read_region` returns, on the GPU) -> u8 [n * k * k, tile_px, tile_px, 3]: resize to k * tile_px, drop alpha,
    crop -- tiles of supertile i are rows i*k*k .. (i+1)*k*k - 1 in (y outer, x inner) order."""
    if not rgba.is_cuda:
        raise RuntimeError("supertiles_to_tiles needs the supertiles on the GPU (no CPU fallback)")
    if rgba.dtype != torch.uint8 or rgba.dim() != 4 or rgba.shape[-1] != 4 or rgba.shape[1] != rgba.shape[2]:
        raise ValueError(f"expected u8 [n, S, S, 4], got {rgba.dtype} {tuple(rgba.shape)}")
    rgba = rgba.contiguous()
    n, S = rgba.shape[0], rgba.shape[1]
    k, t = int(tiles_per_side), int(tile_size_px)
    dev = rgba.device
    b_d, t_d, ksize = _resize_tables_on_device(S, k * t, dev.index if dev.index is not None else torch.cuda.current_device())
    lib = _lib.lib()
    nb = lib.amds_supertiles_to_tiles_workspace_bytes(n, S, k, t)
    if out is None:
        out = torch.empty(n * k * k, t, t, 3, dtype=torch.uint8, device=dev)
    else:           # caller-owned destination (a pipeline's ring buffer): the first n*k*k tiles of it
        if out.dtype != torch.uint8 or not out.is_contiguous() or out.shape[0] < n * k * k or tuple(out.shape[1:]) != (t, t, 3):
            raise ValueError(f"out must be contiguous u8 [>= {n * k * k}, {t}, {t}, 3], got {tuple(out.shape)}")
        out = out[: n * k * k]
    ws = workspace if workspace is not None and workspace.numel() >= nb else torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
    _lib.check(lib.amds_supertiles_to_tiles_u8(rgba.data_ptr(), out.data_ptr(), n, S, k, t, b_d.data_ptr(), t_d.data_ptr(), ksize, ws.data_ptr(), nb,
                                               ops._stream()), "supertiles_to_tiles")
    return out
