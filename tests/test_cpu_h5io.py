"""STAMP's HDF5 feature-file schema through stamp_amd.h5io (libhdf5 via ctypes when h5py is absent): round trips, the three
coordinate conventions of the reference's `get_coords`, `detect_feature_type`'s rule, atomic replace."""
import numpy as np
import pytest

h5io = pytest.importorskip("stamp_amd.h5io")


def _have_backend():
    if h5io._h5py is not None:
        return True
    try:
        h5io._lib()
        return True
    except RuntimeError:
        return False


pytestmark = pytest.mark.skipif(not _have_backend(), reason="neither h5py nor libhdf5 on this machine")


def test_tile_feature_file_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((37, 1024)).astype(np.float16)
    coords = (rng.integers(0, 300, (37, 2)) * 256.0).astype(np.float32)
    p = tmp_path / "sub" / "slide.h5"
    h5io.write_tile_features(p, feats, coords, extractor="uni2", tile_size_um=256.0, tile_size_px=224, code_hash="0a1b2c3d", stamp_version="2.5.0")
    assert p.is_file() and [q.name for q in p.parent.iterdir()] == ["slide.h5"]          # temporary file renamed, nothing left behind
    d, a = h5io.read_file(p)
    assert d["feats"].dtype == np.float16 and np.array_equal(d["feats"].view(np.uint16), feats.view(np.uint16))     # fp16 bit patterns preserved
    assert d["coords"].dtype == np.float32 and np.array_equal(d["coords"], coords)
    assert a == {"stamp_version": "2.5.0", "extractor": "uni2", "unit": "um", "tile_size_um": 256.0, "tile_size_px": 224, "code_hash": "0a1b2c3d", "feat_type": "tile"}
    assert isinstance(a["tile_size_um"], float) and isinstance(a["tile_size_px"], int) and isinstance(a["extractor"], str)
    f2, ci, _ = h5io.read_tile_features(p)
    assert np.array_equal(f2, feats) and ci.tile_size_um == 256.0 and ci.tile_size_px == 224 and abs(ci.mpp - 256 / 224) < 1e-12
    assert h5io.feature_type(a) == "tile"
    # torch tensors are accepted, empty slides too
    import torch
    h5io.write_tile_features(tmp_path / "t.h5", torch.from_numpy(feats), torch.from_numpy(coords), extractor="x", tile_size_um=256.0, tile_size_px=224,
                             code_hash="h", stamp_version="2.5.0")
    assert np.array_equal(h5io.read_file(tmp_path / "t.h5")[0]["feats"], feats)
    with pytest.raises(ValueError):
        h5io.write_tile_features(tmp_path / "bad.h5", feats, coords[:5], extractor="x", tile_size_um=1.0, tile_size_px=1, code_hash="h", stamp_version="v")
    assert not (tmp_path / "bad.h5").exists()


def test_slide_feature_file_and_feature_type(tmp_path):
    v = np.arange(768, dtype=np.float32) / 7
    h5io.write_slide_features(tmp_path / "p.h5", v, encoder="chief", precision="torch.float32", code_hash="deadbeef", stamp_version="2.5.0", feat_type="patient")
    d, a = h5io.read_file(tmp_path / "p.h5")
    assert np.array_equal(d["feats"], v) and a["encoder"] == "chief" and a["version"] == "2.5.0" and a["precision"] == "torch.float32"
    assert h5io.feature_type(a) == "patient"
    assert h5io.feature_type({}) == "tile" and h5io.feature_type({"encoder": "chief"}) == "None"        # str(None): the reference's own quirk (data.py:443-444)
    with pytest.raises(ValueError):
        h5io.write_slide_features(tmp_path / "q.h5", v, encoder="e", precision="p", code_hash="c", stamp_version="v", feat_type="tile")


def test_get_coords_conventions():
    """reference src/stamp/modeling/data.py:741-808."""
    grid = np.stack(np.meshgrid(np.arange(4), np.arange(3)), -1).reshape(-1, 2).astype(np.float32)
    # current format
    ci = h5io.get_coords({"coords": grid * 256.0}, {"tile_size_um": 256.0, "tile_size_px": 224})
    assert np.array_equal(ci.coords_um, grid * 256.0) and ci.tile_size_um == 256.0 and ci.tile_size_px == 224
    # STAMP v2 format: tile_size + unit
    ci = h5io.get_coords({"coords": grid * 112.0}, {"tile_size": 112.0, "unit": "um"})
    assert ci.tile_size_um == 112.0 and ci.tile_size_px is None
    with pytest.raises(RuntimeError):
        _ = ci.mpp
    # historic format: pixel coordinates with a stride of 224 -> 256 um / 224 px
    ci = h5io.get_coords({"coords": grid * 224.0}, {})
    assert ci.tile_size_um == 256.0 and ci.tile_size_px == 224 and np.allclose(ci.coords_um, grid * 256.0)
    # nothing to go by
    with pytest.raises(RuntimeError):
        h5io.get_coords({"coords": grid * 100.0}, {})
    # multiplex bypass: no coords, fake ones from the row count
    ci = h5io.get_coords({"patch_embeddings": np.zeros((5, 8), np.float32)}, {})
    assert np.array_equal(ci.coords_um, np.stack([np.arange(5), np.zeros(5)], 1).astype(np.float32)) and ci.tile_size_um == 0.0
