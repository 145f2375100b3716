"""Oracle for the WSI tiling path ("next" row N3) -- numpy restatement; TEST INFRASTRUCTURE ONLY (tests/,
smoke() and bench.py's cpu_baseline leg may import it, nothing under stamp_amd/ does).

Restates reference src/stamp/preprocessing/tiling.py: `_supertiles` :294-347 (geometry :304-317, `read_region(...).resize((S', S'))
.convert("RGB")` :326-343), `_foreground_coords` :250-277, `_tiles` :196-247 (crop order and micrometre coordinates :225-246).
The resize itself is third-party arithmetic: Pillow (reference uv.lock pins 12.1.1; 12.2.0 is installed here).  Its published
algorithm is restated -- `Image.resize` on "RGBA" = convert to premultiplied "RGBa", `ImagingResample` (two passes, horizontal
first, 8-bit intermediate, taps normalised then rounded to 22-bit fixed point, bicubic a = -0.5), convert back -- and PINNED bit
for bit against the installed Pillow (tests/test_oracle_tiling.py) and, through tests/golden/tiling_*.npz, against the
reference's own functions run on a synthetic slide object.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc: (bounds [out][2], fixed-point taps [out][ksize])."""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            k[x] = _bicubic((x + xmin - center + 0.5) * ss)
            ww += k[x]
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(k[x] * (1 << PRECISION_BITS) - 0.5) if k[x] < 0 else int(k[x] * (1 << PRECISION_BITS) + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, kk = precompute_coeffs(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img_u8: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL `Image.resize((out_w, out_h))` (bicubic) of an [H, W, C] uint8 image without alpha handling."""
    t = _resample_axis(img_u8, out_w, 1) if img_u8.shape[1] != out_w else img_u8
    return _resample_axis(t, out_h, 0) if img_u8.shape[0] != out_h else t


def premultiply(rgba: np.ndarray) -> np.ndarray:
    t = rgba[..., :3].astype(np.int64) * rgba[..., 3:4].astype(np.int64) + 128
    return np.concatenate([(((t >> 8) + t) >> 8).astype(np.uint8), rgba[..., 3:4]], axis=-1)


def unpremultiply_rgb(rgba_p: np.ndarray) -> np.ndarray:
    al = rgba_p[..., 3:4].astype(np.int64)
    c = rgba_p[..., :3].astype(np.int64)
    return np.where((al == 255) | (al == 0), c, np.clip((255 * c) // np.maximum(al, 1), 0, 255)).astype(np.uint8)


def supertile_to_rgb(rgba_u8: np.ndarray, out_px: int) -> np.ndarray:
    """`read_region(...)` result [S, S, 4] -> `.resize((out_px, out_px)).convert("RGB")` [out_px, out_px, 3]  (tiling.py:328-335)."""
    return unpremultiply_rgb(pil_resize_bicubic(premultiply(rgba_u8), out_px, out_px))


def supertile_geometry(slide_mpp: float, tile_size_um: float, tile_size_px: int, max_supertile_size_slide_px: int):
    """tiling.py:304-317 -> (tile_size_slide_px, tiles per side, supertile_size_slide_px, supertile_size_tile_px, supertile_size_um)."""
    max_supertile_um = max_supertile_size_slide_px * slide_mpp
    k = max(int(max_supertile_um // tile_size_um), 1)
    tile_size_slide_px = int(np.ceil(tile_size_um / slide_mpp))
    s_slide = tile_size_slide_px * k
    return tile_size_slide_px, k, s_slide, tile_size_px * k, s_slide * slide_mpp


def foreground_cells(dimensions, thumb_gray_i32: np.ndarray, supertile_slide_px: int, brightness_cutoff):
    """tiling.py:264-277 given the [grid_h, grid_w] int32 luma thumbnail: slide-pixel origins of the supertiles to read, row-major."""
    fg = thumb_gray_i32 < brightness_cutoff if brightness_cutoff is not None else np.ones_like(thumb_gray_i32, dtype=bool)
    out = []
    for y in range(0, dimensions[1], supertile_slide_px):
        for x in range(0, dimensions[0], supertile_slide_px):
            if fg[y // supertile_slide_px, x // supertile_slide_px]:
                out.append((x, y))
    return out


def luma_i(rgb_u8: np.ndarray) -> np.ndarray:
    """PIL convert("I") of RGB (Convert.c rgb2i: L24 >> 16; ITU-R 601 weights in 16-bit fixed point)."""
    r, g, b = (rgb_u8[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.int32)


def tiles_of_supertile(rgb: np.ndarray, origin_um, tile_size_um: float, tile_size_px: int):
    """tiling.py:225-246: crops in (y outer, x inner) order with their top-left micrometre coordinates."""
    n = rgb.shape[0] // tile_size_px
    out = []
    for y in range(n):
        for x in range(n):
            out.append((rgb[y * tile_size_px:(y + 1) * tile_size_px, x * tile_size_px:(x + 1) * tile_size_px],
                        (origin_um[0] + x * tile_size_um, origin_um[1] + y * tile_size_um)))
    return out


def synthetic_slide(width: int, height: int, seed: int) -> np.ndarray:
    """Level-0 pixels of a synthetic slide, RGB u8 [height, width, 3]: white background with a few H&E-coloured textured blobs.
    Deterministic in (width, height, seed); tests regenerate it instead of storing it."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.full((height, width, 3), 244.0, dtype=np.float32)
    for _ in range(5):
        cx, cy, r = rng.uniform(0, width), rng.uniform(0, height), rng.uniform(0.12, 0.3) * min(width, height)
        d = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
        m = np.clip((r - d) / (0.15 * r), 0, 1)[..., None]
        tex = 0.5 + 0.5 * np.sin(xx / rng.uniform(3, 9) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(3, 9))
        col = np.array([200, 120, 180], np.float32) * (1 - tex[..., None]) + np.array([110, 60, 150], np.float32) * tex[..., None]
        img = img * (1 - m) + col * m
    img += rng.normal(0, 3, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)
