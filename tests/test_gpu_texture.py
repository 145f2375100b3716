"""HIP tile background filter (grey + Canny + edge fraction) against the oracle: bit-exact."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import texture
from stamp_amd import ops

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _check(tiles_np, gpu, low=40, high=100):
    frac, edges, gray = ops.tile_edge_fraction(torch.from_numpy(tiles_np).to(gpu), low, high, return_maps=True)
    for i, t in enumerate(tiles_np):
        g = texture.gray_L(t)
        e = texture.canny_l1(g, low, high)
        assert np.array_equal(gray[i].cpu().numpy(), g)
        assert np.array_equal(edges[i].cpu().numpy(), e), f"tile {i}: {(edges[i].cpu().numpy() != e).sum()} pixels differ"
        assert frac[i].item() == pytest.approx(e.mean() / 255, abs=1e-7)
    return frac.cpu().numpy()


def test_gray_matches_pillow_golden(gpu):
    z = np.load(G / "texture_gray.npz")
    _, _, gray = ops.tile_edge_fraction(torch.from_numpy(z["tiles"]).to(gpu), return_maps=True)
    assert np.array_equal(gray.cpu().numpy(), z["gray"])


def test_edge_maps_bit_exact(gpu):
    z = np.load(G / "texture_gray.npz")
    f = _check(z["tiles"], gpu)
    assert f[0] > 0.3 and f[1] < 0.05           # noise is all edges, the smooth H&E-like tile nearly none
    rng = np.random.default_rng(0)
    # blurred noise at several contrasts: long weak chains -> many hysteresis rounds
    base = rng.normal(size=(6, 224, 224))
    from scipy import ndimage
    sm = np.stack([ndimage.gaussian_filter(b, s) for b, s in zip(base, (1, 2, 3, 4, 6, 8))])
    sm = (sm - sm.min((1, 2), keepdims=True)) / (np.ptp(sm, axis=(1, 2), keepdims=True))
    tiles = (np.stack([sm, sm ** 2, 1 - sm], -1) * np.array([255, 200, 120])).astype(np.uint8)
    _check(tiles, gpu)
    _check(tiles, gpu, low=10, high=200)


@pytest.mark.parametrize("S", [3, 17, 64, 112])
def test_small_and_odd_sizes(gpu, S):
    rng = np.random.default_rng(S)
    tiles = rng.integers(0, 256, size=(3, S, S, 3), dtype=np.uint8)
    tiles[1] = 0
    tiles[2, :, S // 2:] = 255
    tiles[2, :, :S // 2] = 0
    _check(tiles, gpu)


def test_filter_decision_and_errors(gpu):
    from stamp_amd.extractor import has_enough_texture
    z = np.load(G / "texture_gray.npz")
    keep = has_enough_texture(torch.from_numpy(z["tiles"]).to(gpu), cutoff=0.02)
    assert keep.tolist() == [texture.has_enough_texture(t, 0.02) for t in z["tiles"]]
    with pytest.raises(RuntimeError):
        ops.tile_edge_fraction(torch.zeros(1, 300, 300, 3, dtype=torch.uint8, device=gpu))
    assert ops.tile_edge_fraction(torch.zeros(0, 224, 224, 3, dtype=torch.uint8, device=gpu)).shape == (0,)
