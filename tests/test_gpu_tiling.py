"""Supertile -> tiles on the GPU (amds_supertiles_to_tiles_u8) against the reference's own tiles (tests/golden/tiling_*.npz, produced by
the reference's `_supertiles` / `_tiles` on a synthetic slide) and against Pillow-pinned oracle resizes: BIT-EXACT."""
import zlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import tiling as ot
from stamp_amd import tiling as pt

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _read_region(slide_rgb, x, y, s):
    out = np.zeros((s, s, 4), dtype=np.uint8)
    sub = slide_rgb[y:y + s, x:x + s]
    out[:sub.shape[0], :sub.shape[1], :3] = sub
    out[:sub.shape[0], :sub.shape[1], 3] = 255
    return out


@pytest.mark.parametrize("tag", ["mpp050", "mpp025"])
def test_gpu_tiles_equal_the_references(gpu, tag):
    z = np.load(G / f"tiling_{tag}.npz")
    w, h, seed = (int(v) for v in z["slide"])
    mpp = float(z["mpp"])
    slide = ot.synthetic_slide(w, h, seed)
    geo = pt.supertile_geometry(mpp, 256.0, 224, 1024)
    fg = [tuple(int(v) for v in r) for r in z["foreground"]]
    regions = np.stack([_read_region(slide, x, y, geo.supertile_size_slide_px) for x, y in fg])           # what openslide.read_region hands over
    tiles = pt.supertiles_to_tiles(torch.from_numpy(regions).to(gpu), geo.tiles_per_side, 224).cpu().numpy()
    coords = np.concatenate([pt.tile_coords_um(o, mpp, geo.tiles_per_side, 256.0) for o in fg])
    order = np.lexsort((coords[:, 0], coords[:, 1]))
    assert np.array_equal(coords[order], z["coords_um"])
    crc = np.array([zlib.crc32(tiles[i].tobytes()) for i in order], dtype=np.uint32)
    assert np.array_equal(crc, z["crc32"])                                  # every tile, every byte, equals the reference's PIL pipeline
    for i, ref in zip(z["full_idx"], z["full_tiles"]):
        assert np.array_equal(tiles[order[int(i)]], ref)


@pytest.mark.parametrize("S,k", [(1024, 1), (1024, 2), (512, 1), (896, 4), (100, 1)])
def test_gpu_resize_random_rgba_vs_oracle(gpu, S, k):
    """Random pixels incl. partial transparency (premultiply / un-premultiply paths), down- and up-scaling."""
    rng = np.random.default_rng(S + k)
    a = rng.integers(0, 256, (3, S, S, 4), dtype=np.uint8)
    a[0, :, :, 3] = 255
    a[1, : S // 2, :, 3] = 255
    a[1, S // 2:, :, :] = 0
    t = 224 if S >= 512 else 64
    got = pt.supertiles_to_tiles(torch.from_numpy(a).to(gpu), k, t).cpu().numpy()
    assert got.shape == (3 * k * k, t, t, 3)
    for i in range(3):
        rgb = ot.supertile_to_rgb(a[i], k * t)
        ref = np.stack([tile for tile, _ in ot.tiles_of_supertile(rgb, (0.0, 0.0), 1.0, t)])
        assert np.array_equal(got[i * k * k:(i + 1) * k * k], ref), (S, k, i)
    assert pt.supertiles_to_tiles(torch.from_numpy(a[:0]).to(gpu), k, t).shape == (0, t, t, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        pt.supertiles_to_tiles(torch.from_numpy(a), k, t)
