"""Oracle for Macenko stain normalisation (BASELINE.json north_star names it; SURVEY.md F1: the reference itself does NOT contain it).

PARITY STATUS: **parity unpinned** -- there is no reference implementation, test or golden vector for this step in KatherLab/STAMP
v2.5.0.  What is restated is the published algorithm (M. Macenko et al., "A method for normalizing histology slides for quantitative
analysis", ISBI 2009) in the form its widely used numpy transcription takes: optical density OD = -log((I + 1) / Io); pixels with any
OD channel < beta are transparent and ignored for the fit; the plane of the two largest eigenvectors of cov(OD); the alpha / (100 -
alpha) percentiles of the angle in that plane give the two stain vectors (haematoxylin = the one with the larger red OD); stain
concentrations by least squares; the 99th percentile of each concentration is mapped onto reference maxima; the image is rebuilt with
reference stain vectors.  Conventions fixed here (the paper leaves them open): the largest eigenvector is oriented so that its
components sum to a positive number (all projections on it are then positive and the angle lives in (0, pi): no wrap-around);
percentiles are numpy's linear-interpolation percentiles; output = clip(floor(Io * exp(-HERef C)), 0, 255).
Test infrastructure only.
"""
from __future__ import annotations

import numpy as np

HE_REF = np.array([[0.5626, 0.2159], [0.7201, 0.8012], [0.4062, 0.5581]])
MAXC_REF = np.array([1.9705, 1.0308])


def macenko_fit(tile_u8: np.ndarray, Io: float = 240.0, alpha: float = 1.0, beta: float = 0.15):
    """-> (HE [3, 2] stain vectors, maxC [2]) of one RGB tile, or None if fewer than 16 pixels are stained."""
    od = -np.log((tile_u8.reshape(-1, 3).astype(np.float64) + 1.0) / Io)
    keep = ~np.any(od < beta, axis=1)
    if keep.sum() < 16:
        return None
    odh = od[keep]
    w, v = np.linalg.eigh(np.cov(odh.T))
    e1, e2 = v[:, 1].copy(), v[:, 2].copy()                   # second largest, largest
    if e2.sum() < 0:
        e2 = -e2
    that = odh @ np.stack([e1, e2], axis=1)
    phi = np.arctan2(that[:, 1], that[:, 0])
    lo, hi = np.percentile(phi, alpha), np.percentile(phi, 100 - alpha)
    vmin = e1 * np.cos(lo) + e2 * np.sin(lo)
    vmax = e1 * np.cos(hi) + e2 * np.sin(hi)
    he = np.stack([vmin, vmax], axis=1) if vmin[0] > vmax[0] else np.stack([vmax, vmin], axis=1)
    c = np.linalg.lstsq(he, od.T, rcond=None)[0]
    return he, np.array([np.percentile(c[0], 99), np.percentile(c[1], 99)])


def macenko_normalize(tile_u8: np.ndarray, Io: float = 240.0, alpha: float = 1.0, beta: float = 0.15) -> np.ndarray:
    fit = macenko_fit(tile_u8, Io, alpha, beta)
    if fit is None:
        return tile_u8.copy()                                 # (almost) no stained pixel: background tile, passed through
    he, maxc = fit
    od = -np.log((tile_u8.reshape(-1, 3).astype(np.float64) + 1.0) / Io)
    c = np.linalg.lstsq(he, od.T, rcond=None)[0]
    c2 = c * (MAXC_REF / maxc)[:, None]
    out = Io * np.exp(-HE_REF @ c2)
    return np.clip(np.floor(out), 0, 255).T.reshape(tile_u8.shape).astype(np.uint8)


def synthetic_he_tiles(n: int, size: int, seed: int, he=None) -> np.ndarray:
    """Tiles synthesised FROM known stain vectors (Beer-Lambert: I = Io exp(-HE C) with smooth random concentration fields + sensor noise),
    so that a fit has a known answer."""
    rng = np.random.default_rng(seed)
    he = np.array([[0.65, 0.07], [0.70, 0.99], [0.29, 0.11]]) if he is None else np.asarray(he)
    he = he / np.linalg.norm(he, axis=0, keepdims=True)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64) / size
    out = np.empty((n, size, size, 3), np.uint8)
    for i in range(n):
        f, ph = rng.uniform(1.0, 6.0, size=4), rng.uniform(0, 6.28, size=4)
        a = 0.5 + 0.5 * np.sin(6.28 * f[0] * xx + ph[0]) * np.cos(6.28 * f[1] * yy + ph[1])
        b = 0.5 + 0.5 * np.sin(6.28 * f[2] * (xx + yy) + ph[2]) * np.cos(6.28 * f[3] * (xx - yy) + ph[3])
        a, b = np.clip(1.6 * a - 0.3, 0, None), np.clip(1.6 * b - 0.3, 0, None)          # regions of (almost) pure stain exist
        od = (1.4 * a)[..., None] * he[:, 0] + (1.0 * b)[..., None] * he[:, 1]
        out[i] = np.clip(240.0 * np.exp(-od) + rng.normal(0, 1.5, size=od.shape), 0, 255).astype(np.uint8)
    return out
