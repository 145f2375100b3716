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


def test_written_stamp_version_is_what_the_reference_can_parse(tmp_path):
    """The reference does `Version(attrs["stamp_version"]) > Version(stamp.__version__)` on every feature file it opens
    (modeling/data.py:793-799; stamp 2.5.0, pyproject.toml:3): what THIS package writes by default must parse and must not be newer."""
    from packaging.version import Version

    from stamp_amd import encoder
    assert Version(encoder.STAMP_FORMAT_VERSION) <= Version("2.5.0")
    assert encoder.VERSION == encoder.STAMP_FORMAT_VERSION
    enc = encoder.HipGatedAttentionEncoder.__new__(encoder.HipGatedAttentionEncoder)
    enc.identifier, enc.precision = "chief", "torch.float32"
    enc._save_features_(tmp_path / "s.h5", np.zeros(8, np.float32), "slide")
    a = h5io.read_file(tmp_path / "s.h5")[1]
    assert Version(a["stamp_version"]) <= Version("2.5.0") and Version(a["version"]) <= Version("2.5.0")
    assert a["amdstamp_version"] == encoder.AMDSTAMP_VERSION
    # the writers refuse a string the reference would crash on
    with pytest.raises(ValueError, match="PEP 440"):
        h5io.write_slide_features(tmp_path / "q.h5", np.zeros(8, np.float32), encoder="e", precision="p", code_hash="c", stamp_version="amdstamp-0.2")
    with pytest.raises(ValueError, match="PEP 440"):
        h5io.write_tile_features(tmp_path / "r.h5", np.zeros((1, 8), np.float16), np.zeros((1, 2), np.float32), extractor="x", tile_size_um=1.0,
                                 tile_size_px=1, code_hash="h", stamp_version="amdstamp-0.2")


def test_attribute_types_on_disk_are_h5pys(tmp_path):
    """h5py stores a Python str attribute as a scalar variable-length UTF-8 string, a Python float as IEEE f64 and a Python int as i64
    (what `h5file.attrs[...] = value` does in preprocessing/__init__.py:350-359).  Read the file back with the HDF5 command-line tool, an
    independent reader, instead of with h5io itself."""
    import shutil
    import subprocess
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if __import__("os").path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is None:
        pytest.skip("no h5dump on this machine")
    feats = np.arange(24, dtype=np.float16).reshape(3, 8)
    h5io.write_tile_features(tmp_path / "t.h5", feats, np.ones((3, 2), np.float32), extractor="uni2", tile_size_um=256.0, tile_size_px=224,
                             code_hash="0a1b2c3d", stamp_version="2.5.0")
    out = subprocess.run([h5dump, "-H", "-A", str(tmp_path / "t.h5")], capture_output=True, text=True, check=True).stdout
    blocks = {}
    for chunk in out.split('ATTRIBUTE "')[1:]:
        blocks[chunk.split('"', 1)[0]] = chunk
    for k in ("stamp_version", "extractor", "unit", "code_hash", "feat_type"):
        b = blocks[k]
        assert "H5T_STRING" in b and "STRSIZE H5T_VARIABLE" in b and "CSET H5T_CSET_UTF8" in b and "DATASPACE  SCALAR" in b, (k, b)
    assert "H5T_IEEE_F64LE" in blocks["tile_size_um"] and "DATASPACE  SCALAR" in blocks["tile_size_um"]
    assert "H5T_STD_I64LE" in blocks["tile_size_px"] and "DATASPACE  SCALAR" in blocks["tile_size_px"]
    full = subprocess.run([h5dump, "-H", str(tmp_path / "t.h5")], capture_output=True, text=True, check=True).stdout
    d = {c.split('"', 1)[0]: c for c in full.split('DATASET "')[1:]}
    assert "H5T_IEEE_F16LE" in d["feats"] or "16-bit little-endian floating-point" in d["feats"], d["feats"]     # h5dump 1.10 has no name for f16
    assert "( 3, 8 )" in d["feats"] and "H5T_IEEE_F32LE" in d["coords"] and "( 3, 2 )" in d["coords"]
