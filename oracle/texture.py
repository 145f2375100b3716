"""Oracle for the tile background filter: `_has_enough_texture`,
reference src/stamp/preprocessing/tiling.py:280-291:  tile.convert("L") -> cv2.Canny(gray, 40, 100) -> mean/255 >= cutoff.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY STATUS
* grey conversion: **pinned** -- `PIL.Image.convert("L")` (Pillow 12.x is installed here; the reference's lock file has
  pillow 11/12) produced tests/golden/texture_gray.npz via tools/make_golden.py; the restatement is Pillow's fixed-point
  ITU-R 601 rule  L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16.
* Canny: **parity unpinned** -- the arithmetic lives in `opencv-python==4.13.0.92` (reference uv.lock), which is neither
  under /root/reference nor installed here, and no reference test holds an edge map (the only pin is the end-to-end
  network golden of tests/test_feature_extractors.py:84-169).  Restated from OpenCV's published algorithm for
  `Canny(image, threshold1, threshold2)` with the defaults apertureSize = 3, L2gradient = false:
    Sobel 3x3 derivatives with BORDER_REPLICATE (int16), magnitude |dx| + |dy|, low = floor(threshold1), high = floor(threshold2);
    non-maximum suppression on pixels with mag > low using the fixed-point sector test (TG22 = round(tan(22.5deg) * 2^15) =
    13573): horizontal if |dy| << 15 < |dx| * TG22 (keep if mag > left and mag >= right), vertical if |dy| << 15 >
    |dx| * TG22 + (|dx| << 16) (keep if mag > up and mag >= down), else diagonal with s = sign(dx ^ dy) (keep if mag >
    both diagonal neighbours); the magnitude plane is 0 outside the image; survivors with mag > high are edges, the others
    candidates; candidates 8-connected to an edge become edges; output 255 on edges.
  Written here with whole-array numpy operations and scipy's connected-component labelling, i.e. independently of the
  per-pixel / relaxation form of the HIP kernel, so the two at least check each other.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def gray_L(rgb_u8: np.ndarray) -> np.ndarray:
    """[..., 3] u8 -> [...] u8, Pillow `convert("L")`."""
    r, g, b = (rgb_u8[..., i].astype(np.int64) for i in range(3))
    return ((19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16).astype(np.uint8)


def canny_l1(gray_u8: np.ndarray, low: int = 40, high: int = 100) -> np.ndarray:
    """[S, S] u8 -> [S, S] u8 edge map (0 / 255)."""
    g = np.pad(gray_u8.astype(np.int32), 1, mode="edge")
    a, b, c = g[:-2, :-2], g[:-2, 1:-1], g[:-2, 2:]
    d, f = g[1:-1, :-2], g[1:-1, 2:]
    p, q, r = g[2:, :-2], g[2:, 1:-1], g[2:, 2:]
    dx = (c + 2 * f + r) - (a + 2 * d + p)
    dy = (p + 2 * q + r) - (a + 2 * b + c)
    mag = np.abs(dx) + np.abs(dy)
    mp = np.pad(mag, 1)                                   # zero outside the image
    H, W = mag.shape
    yy, xx = np.mgrid[0:H, 0:W]

    def at(oy, ox):
        return mp[yy + 1 + oy, xx + 1 + ox]

    ax, ay = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ~horiz & (ay > tg67)
    diag = ~horiz & ~vert
    s = np.where((dx ^ dy) < 0, -1, 1)
    keep_h = (mag > at(0, -1)) & (mag >= at(0, 1))
    keep_v = (mag > at(-1, 0)) & (mag >= at(1, 0))
    up = np.where(s > 0, at(-1, -1), at(-1, 1))           # mag[y-1][x-s]
    dn = np.where(s > 0, at(1, 1), at(1, -1))             # mag[y+1][x+s]
    keep_d = (mag > up) & (mag > dn)
    keep = (mag > low) & ((horiz & keep_h) | (vert & keep_v) | (diag & keep_d))
    strong = keep & (mag > high)
    lab, n = ndimage.label(keep, structure=np.ones((3, 3), dtype=bool))
    if n == 0:
        return np.zeros_like(gray_u8)
    has_strong = np.zeros(n + 1, dtype=bool)
    has_strong[np.unique(lab[strong])] = True
    has_strong[0] = False
    return np.where(has_strong[lab], 255, 0).astype(np.uint8)


def edge_fraction(tile_rgb_u8: np.ndarray, low: int = 40, high: int = 100) -> float:
    """tiling.py:286-287: `np.array(edges).mean() / 255`."""
    return float(canny_l1(gray_L(tile_rgb_u8), low, high).mean() / 255)


def has_enough_texture(tile_rgb_u8: np.ndarray, cutoff: float) -> bool:
    return bool(edge_fraction(tile_rgb_u8) >= cutoff)
