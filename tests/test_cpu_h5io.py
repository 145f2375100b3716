"""STAMP's HDF5 feature-file schema through stamp_amd.h5io (h5py, else libhdf5 via ctypes, else the pure-Python subset): round trips, the three
coordinate conventions of the reference's `get_coords`, `detect_feature_type`'s rule, atomic replace."""
from pathlib import Path

import numpy as np
import pytest

h5io = pytest.importorskip("stamp_amd.h5io")      # (every machine has a backend: h5py, libhdf5 via ctypes, or stamp_amd/h5min.py)


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


# ---- the pure-Python backend (stamp_amd/h5min.py): machines with neither h5py nor libhdf5 ------------------------------------------------
def _have_c():
    try:
        h5io._lib()
        return True
    except RuntimeError:
        return False


_ATTRS = {"stamp_version": "2.5.0", "extractor": "virchow2-é", "unit": "um", "tile_size_um": 256.0, "tile_size_px": 224, "code_hash": "0a1b2c3d",
          "feat_type": "tile"}


def test_h5min_round_trip_and_edge_shapes(tmp_path):
    from stamp_amd import h5min
    rng = np.random.default_rng(0)
    for n in (37, 1, 0):
        feats = rng.standard_normal((n, 1024)).astype(np.float16)
        coords = (rng.random((n, 2)) * 1e4).astype(np.float32)
        p = tmp_path / f"t{n}.h5"
        h5min.write(p, {"feats": feats, "coords": coords}, _ATTRS)
        d, a = h5min.read(p)
        assert a == _ATTRS and list(a) == list(_ATTRS)
        assert d["feats"].dtype == np.float16 and d["feats"].shape == (n, 1024) and np.array_equal(d["feats"], feats)
        assert d["coords"].dtype == np.float32 and np.array_equal(d["coords"], coords)
    h5min.write(tmp_path / "s.h5", {"feats": np.arange(768, dtype=np.float32)}, {"encoder": "chief", "flag": True, "n": np.int64(-3), "x": np.float32(0.5)})
    d, a = h5min.read(tmp_path / "s.h5", want=("feats", "coords"))
    assert set(d) == {"feats"} and d["feats"].shape == (768,) and a == {"encoder": "chief", "flag": 1, "n": -3, "x": 0.5}
    with pytest.raises(ValueError, match="not an HDF5 file"):
        (tmp_path / "junk.h5").write_bytes(b"x" * 200)
        h5min.read(tmp_path / "junk.h5")
    raw = bytearray((tmp_path / "s.h5").read_bytes())
    raw[8] = 2                                              # superblock version 2: the "latest" format family
    (tmp_path / "v2.h5").write_bytes(bytes(raw))
    with pytest.raises(h5min.Unsupported, match="superblock version 2"):
        h5min.read(tmp_path / "v2.h5")
    with pytest.raises(TypeError):
        h5min.write(tmp_path / "bad.h5", {}, {"a": [1, 2]})


def test_h5min_files_interchange_with_libhdf5(tmp_path, monkeypatch):
    """Both directions against the real library (what h5py wraps), where it exists: a file written by the pure-Python backend reads back
    identically through libhdf5 and h5dump prints the same types and values for it as for the library's own file; a file written by
    libhdf5 reads back identically through the pure-Python reader."""
    from stamp_amd import h5min
    if not _have_c():
        pytest.skip("no libhdf5 on this machine to compare with")
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((211, 1536)).astype(np.float16)
    coords = (rng.random((211, 2)) * 1e5).astype(np.float32)
    pm, pc = tmp_path / "m.h5", tmp_path / "c.h5"
    h5min.write(pm, {"feats": feats, "coords": coords}, _ATTRS)
    d, a = h5io._c_read(str(pm))
    assert np.array_equal(d["feats"], feats) and d["feats"].dtype == np.float16 and np.array_equal(d["coords"], coords) and a == _ATTRS
    monkeypatch.setenv("AMDSTAMP_H5_BACKEND", "c")
    assert h5io.backend() == "c"
    h5io._write(pc, {"feats": feats, "coords": coords}, _ATTRS)
    d, a = h5min.read(pc)
    assert np.array_equal(d["feats"], feats) and np.array_equal(d["coords"], coords) and a == _ATTRS
    import shutil
    import subprocess
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if Path("/opt/conda/bin/h5dump").exists() else None)
    if h5dump:
        outs = [subprocess.run([h5dump, str(p)], capture_output=True, text=True, check=True).stdout.split("\n", 1)[1] for p in (pm, pc)]
        assert outs[0] == outs[1] and "H5T_VARIABLE" in outs[0] and "H5T_CSET_UTF8" in outs[0]


def test_h5min_reads_chunked_compressed_datasets(tmp_path):
    """Legacy feature files may be chunked and gzip-compressed (h5py `compression="gzip", shuffle=True`): v1 chunk B-tree + deflate /
    shuffle / fletcher32 filters, made here with the real library."""
    import ctypes as C
    from stamp_amd import h5min
    if not _have_c():
        pytest.skip("no libhdf5 on this machine to make the file with")
    lib = h5io._lib()
    hid, u64 = C.c_int64, C.c_uint64
    for name, res, args in (("H5Pcreate", hid, [hid]), ("H5Pset_chunk", C.c_int, [hid, C.c_int, C.POINTER(u64)]), ("H5Pset_deflate", C.c_int, [hid, C.c_uint]),
                            ("H5Pset_shuffle", C.c_int, [hid]), ("H5Pset_fletcher32", C.c_int, [hid]), ("H5Pclose", C.c_int, [hid]),
                            ("H5Zfilter_avail", C.c_int, [C.c_int])):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5Zfilter_avail(1) <= 0:
        pytest.skip("this libhdf5 has no deflate filter")
    rng = np.random.default_rng(2)
    feats = np.round(rng.standard_normal((1000, 96)) * 4).astype(np.float16)            # compressible
    coords = (rng.integers(0, 400, (1000, 2)) * 256).astype(np.float32)
    p = tmp_path / "z.h5"
    f = lib.H5Fcreate(str(p).encode(), 2, 0, 0)
    assert f >= 0
    for nm, arr, chunk in (("feats", feats, (128, 96)), ("coords", coords, (300, 2))):
        dcpl = lib.H5Pcreate(h5io._g("H5P_CLS_DATASET_CREATE_ID_g"))
        assert dcpl >= 0 and lib.H5Pset_chunk(dcpl, 2, (u64 * 2)(*chunk)) >= 0
        assert lib.H5Pset_shuffle(dcpl) >= 0 and lib.H5Pset_deflate(dcpl, 4) >= 0
        if nm == "coords":
            assert lib.H5Pset_fletcher32(dcpl) >= 0
        sp = lib.H5Screate_simple(2, (u64 * 2)(*arr.shape), None)
        t, own = h5io._np_type(lib, arr)
        d = lib.H5Dcreate2(f, nm.encode(), t, sp, 0, dcpl, 0)
        assert d >= 0 and lib.H5Dwrite(d, t, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) >= 0
        lib.H5Dclose(d), lib.H5Sclose(sp), lib.H5Pclose(dcpl)
    lib.H5Fclose(f)
    assert p.stat().st_size < feats.nbytes                                                # it did compress
    d, _ = h5min.read(p)
    assert np.array_equal(d["feats"], feats) and np.array_equal(d["coords"], coords)


def test_public_api_on_the_pure_python_backend(tmp_path, monkeypatch):
    monkeypatch.setenv("AMDSTAMP_H5_BACKEND", "min")
    assert h5io.backend() == "min"
    rng = np.random.default_rng(3)
    feats = rng.standard_normal((50, 768)).astype(np.float16)
    coords = (rng.integers(0, 40, (50, 2)) * 256.0).astype(np.float32)
    p = tmp_path / "a" / "slide.h5"
    h5io.write_tile_features(p, feats, coords, extractor="ctranspath", tile_size_um=256.0, tile_size_px=224, code_hash="deadbeef", stamp_version="2.5.0")
    d, a = h5io.read_file(p)
    assert np.array_equal(d["feats"], feats) and np.array_equal(d["coords"], coords)
    assert a["extractor"] == "ctranspath" and a["feat_type"] == "tile" and a["tile_size_px"] == 224 and a["tile_size_um"] == 256.0 and a["unit"] == "um"
    assert h5io.feature_type(a) == "tile"
    info = h5io.get_coords(d, a)
    assert info.tile_size_um == 256.0 and info.tile_size_px == 224
    monkeypatch.setenv("AMDSTAMP_H5_BACKEND", "nope")
    with pytest.raises(RuntimeError, match="not available"):
        h5io.backend()


def test_get_coords_equals_the_references_own_function():
    """`h5io.get_coords` against results of the reference's own `get_coords` (modeling/data.py:741-808, executed by name over stand-in file objects;
    tests/golden/get_coords.json): STAMP v2, the current format, the historic 224-stride format by attribute and by stride, the coords-less bypass,
    a file from a newer STAMP (refused) and an un-inferable one (refused)."""
    import json

    z = json.loads((Path(__file__).parent / "golden" / "get_coords.json").read_text())
    assert set(z) == {"v2", "current", "historic_attr", "historic_stride", "no_coords", "newer", "unknown"}
    for name, rec in z.items():
        ds = {k: np.asarray(v, dtype=np.float32) for k, v in rec["datasets"].items()}
        if "error" in rec:
            with pytest.raises(RuntimeError) as ei:
                h5io.get_coords(ds, rec["attrs"])
            assert str(ei.value) == rec["message"], name
            continue
        ci = h5io.get_coords(ds, rec["attrs"])
        assert np.array_equal(np.asarray(ci.coords_um, dtype=np.float32), np.asarray(rec["coords_um"], dtype=np.float32)), name
        assert float(ci.tile_size_um) == rec["tile_size_um"] and ci.tile_size_px == rec["tile_size_px"], name


def test_version_checks_without_packaging(monkeypatch):
    """A bare GPU box may lack `packaging`: the PEP 440 check in front of every write and the newer-than comparison in get_coords fall back to
    the PEP's own regular expression and a release-tuple comparison, with the same answers."""
    import builtins

    from packaging.version import InvalidVersion, Version

    cases = ["2.4.0", "2.4.0.dev1", "1.0rc1", "v1.2", "abc", "2.4.0+local.1", "1..2", "2.5", "2.5.0.post1", "1!0.3", ""]
    want = {}
    for v in cases:
        try:
            Version(v)
            want[v] = True
        except InvalidVersion:
            want[v] = False
    real = builtins.__import__

    def no_packaging(name, *a, **k):
        if name.split(".")[0] == "packaging":
            raise ImportError(name)
        return real(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_packaging)
    for v in cases:
        assert h5io._is_pep440(v) == want[v], v
    for a, b in (("2.5.0", "2.4.0"), ("2.4.0", "2.4.0"), ("2.4.0.dev3", "2.4.0"), ("2.4.0.post1", "2.4"), ("2.4", "2.4.0"), ("2.10", "2.9.1")):
        assert h5io._version_newer(a, b) == (Version(a) > Version(b)), (a, b)
    with pytest.raises(ValueError):
        h5io._pep440("not a version")
