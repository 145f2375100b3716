"""STAMP's tile-cache zip format (reference tiling.py:68-168, 380-406) through stamp_amd.tile_cache.  Fixtures from
tools/make_golden.py::golden_tile_cache, which also asserted that the REFERENCE's `_tiles_from_cache_file` reads the zip written here."""
import json
from pathlib import Path
from zipfile import ZipFile

import numpy as np
import pytest

pytest.importorskip("PIL")
from stamp_amd import tile_cache as tc  # noqa: E402

G = Path(__file__).parent / "golden"


def test_reads_reference_style_legacy_cache():
    z = np.load(G / "tile_cache_expect.npz")
    tiles, coords, params = tc.read_tile_cache(G / "tile_cache_ref_style.zip")
    assert "tile_ext" not in params and params["tile_size_um"] == 256.0              # legacy cache: jpg assumed
    assert np.array_equal(coords, z["coords"]) and np.array_equal(tiles, z["jpeg_decoded"])    # same decoder as the reference: identical pixels


def test_round_trip_png_and_names(tmp_path):
    z = np.load(G / "tile_cache_expect.npz")
    tiles, coords, params = tc.read_tile_cache(G / "tile_cache_amd.zip")
    assert np.array_equal(tiles, z["tiles"]) and np.array_equal(coords, z["coords"]) and params["tile_ext"] == "png"
    p = tc.cache_file_path(tmp_path, "/x/y/slide_x.svs", params)
    assert p.name.startswith("slide_x.") and p.name.endswith(".zip") and len(p.name) == len("slide_x.") + 64 + len(".zip")
    assert p == tc.cache_file_path(tmp_path, "/x/y/slide_x.svs", dict(reversed(list(params.items()))))       # sort_keys: order-independent
    tc.write_tile_cache(p, tiles, coords, params)
    with ZipFile(p) as zf:
        names = zf.namelist()
        assert names[0] == "tiler_params.json" and names[3] == "tile_(512.5, 1024.0).png" and json.loads(zf.read(names[0])) == params
    t2, c2, _ = tc.read_tile_cache(p)
    assert np.array_equal(t2, tiles) and np.array_equal(c2, coords)
    with pytest.raises(ValueError):
        tc.write_tile_cache(tmp_path / "bad.zip", tiles, coords, dict(params, tile_ext="bmp"))
    assert not (tmp_path / "bad.zip").exists()
