"""The tiling oracle (oracle/tiling.py) pinned twice: its restatement of Pillow's bicubic resize / premultiplication / luma against the
INSTALLED Pillow, bit for bit, and the whole supertile -> tiles path against fixtures produced by the reference's own `_supertiles`,
`_tiles` and `_foreground_coords` (tools/make_golden.py::golden_tiling).  Then the product's host logic (stamp_amd/tiling.py)."""
import zlib
from pathlib import Path

import numpy as np
import pytest

from oracle import tiling as ot

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

G = Path(__file__).parent / "golden"


@pytest.mark.parametrize("h,w,oh,ow", [(64, 64, 14, 14), (100, 73, 22, 31), (57, 57, 114, 114), (512, 512, 224, 224), (300, 300, 448, 448), (33, 90, 33, 17)])
def test_bicubic_resize_is_pillows_bit_for_bit(h, w, oh, ow):
    rng = np.random.default_rng(h * w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    assert np.array_equal(ot.pil_resize_bicubic(a, ow, oh), np.array(Image.fromarray(a, "RGB").resize((ow, oh))))


def test_rgba_resize_premultiplies_like_pillow():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (96, 96, 4), dtype=np.uint8)
    a[:30, :, 3] = 255
    a[70:, :, 3] = 0
    a[70:, :, :3] = 0                       # openslide: transparent black past the slide edge
    im = Image.fromarray(a, "RGBA")
    assert np.array_equal(ot.premultiply(a), np.array(im.convert("RGBa")))
    assert np.array_equal(ot.supertile_to_rgb(a, 21), np.array(im.resize((21, 21)).convert("RGB")))
    rgb = rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)
    assert np.array_equal(ot.luma_i(rgb), np.array(Image.fromarray(rgb, "RGB").convert("I")))


def _read_region(slide_rgb, x, y, s):
    out = np.zeros((s, s, 4), dtype=np.uint8)
    sub = slide_rgb[y:y + s, x:x + s]
    out[:sub.shape[0], :sub.shape[1], :3] = sub
    out[:sub.shape[0], :sub.shape[1], 3] = 255
    return out


@pytest.mark.parametrize("tag", ["mpp050", "mpp025"])
def test_tiling_matches_reference_fixture(tag):
    """Fixture = the reference's functions on a synthetic slide.  Geometry, background rejection, coordinates and the pixels of EVERY
    tile (CRC32 of the reference's tile images), oracle and product host logic alike."""
    from stamp_amd import tiling as pt

    z = np.load(G / f"tiling_{tag}.npz")
    w, h, seed = (int(v) for v in z["slide"])
    mpp = float(z["mpp"])
    slide = ot.synthetic_slide(w, h, seed)
    tpx, k, s_slide, s_tile, s_um = ot.supertile_geometry(mpp, 256.0, 224, 1024)
    geo = pt.supertile_geometry(mpp, 256.0, 224, 1024)
    assert (geo.tile_size_slide_px, geo.tiles_per_side, geo.supertile_size_slide_px, geo.supertile_size_tile_px, geo.supertile_size_um) == (tpx, k, s_slide, s_tile, s_um)
    assert (k, s_slide) == ((2, 1024) if tag == "mpp050" else (1, 1024))
    thumb = Image.fromarray(z["thumb2x"], "RGB")
    gw, gh = pt.thumbnail_size((w, h), s_slide)
    assert thumb.size == (2 * gw, 2 * gh) or max(thumb.size) <= 2 * max(gw, gh)
    fg = pt.foreground_coords((w, h), thumb, s_slide, 224)
    assert fg == [tuple(int(v) for v in r) for r in z["foreground"]]
    gray = ot.luma_i(np.array(thumb.resize((gw, gh))))
    assert ot.foreground_cells((w, h), gray, s_slide, 224) == fg
    assert len(fg) < gw * gh                                         # something was rejected as background
    tiles = []
    for (x, y) in fg:
        rgb = ot.supertile_to_rgb(_read_region(slide, x, y, s_slide), s_tile)
        got = ot.tiles_of_supertile(rgb, (x * mpp, y * mpp), 256.0, 224)
        cu = pt.tile_coords_um((x, y), mpp, k, 256.0)
        assert np.array_equal(cu, np.array([c for _, c in got]))
        tiles += got
    tiles.sort(key=lambda t: (t[1][1], t[1][0]))
    assert np.array_equal(np.array([c for _, c in tiles]), z["coords_um"])
    crc = np.array([zlib.crc32(np.ascontiguousarray(t).tobytes()) for t, _ in tiles], dtype=np.uint32)
    assert np.array_equal(crc, z["crc32"])
    for i, ref in zip(z["full_idx"], z["full_tiles"]):
        assert np.array_equal(tiles[int(i)][0], ref)


def test_product_coefficients_equal_oracles():
    from stamp_amd import tiling as pt

    for a, b in ((1024, 224), (1024, 448), (512, 224), (100, 224)):
        ob, ok = ot.precompute_coeffs(a, b)
        pb, pk = pt.resize_coefficients(a, b)
        assert np.array_equal(ob, pb) and np.array_equal(ok, pk)


def test_get_slide_mpp_matches_reference_fixture():
    """`tiling.get_slide_mpp` against the reference's own `get_slide_mpp_` (tiling.py:409-475) run on the same property mappings
    (tests/golden/slide_mpp.json, tools/make_golden.py:golden_slide_mpp): every source, their precedence, malformed XML, the default, the error."""
    import json
    import types
    from pathlib import Path

    import pytest

    from stamp_amd.tiling import MPPExtractionError, get_slide_mpp

    cases = json.loads((Path(__file__).parent / "golden" / "slide_mpp.json").read_text())
    assert len(cases) == 16
    for name, rec in cases.items():
        for slide in (types.SimpleNamespace(properties=rec["properties"]), rec["properties"]):          # an opened slide, or its property mapping
            if "error" in rec:
                assert rec["error"] == "MPPExtractionError"
                with pytest.raises(MPPExtractionError):
                    get_slide_mpp(slide, default_mpp=rec["default_mpp"])
            else:
                assert get_slide_mpp(slide, default_mpp=rec["default_mpp"]) == rec["mpp"], name


def test_resolve_slide_mpps_fills_missing_resolutions():
    import types

    from stamp_amd.preprocess import SlideJob, resolve_slide_mpps

    closed = []

    class Fake:
        def __init__(self, props):
            self.properties = props

        def close(self):
            closed.append(self)

    jobs = [SlideJob(lambda: Fake({"openslide.mpp-x": "0.25"}), "a.h5", None, "a"), SlideJob(Fake({}), "b.h5", 0.5, "b"),
            SlideJob(lambda: Fake({}), "c.h5", None, "c"), SlideJob(types.SimpleNamespace(properties={"openslide.comment": "<PixelSizeMicrons>0.4</PixelSizeMicrons>"}), "d.h5", None)]
    known, failed = resolve_slide_mpps(jobs)
    assert [None if j is None else j.slide_mpp for j in known] == [0.25, 0.5, None, 0.4] and list(failed) == [2] and "MPPExtractionError" in failed[2]
    assert len(closed) == 2                                   # the two slides opened here were closed here
    known, failed = resolve_slide_mpps(jobs, default_mpp=1.0)
    assert [j.slide_mpp for j in known] == [0.25, 0.5, 1.0, 0.4] and not failed
