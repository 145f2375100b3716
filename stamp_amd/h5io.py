"""Feature-file I/O in STAMP's HDF5 schema (host side; no arithmetic).

Schema (reference src/stamp/preprocessing/__init__.py:345-367 for tile features, src/stamp/encoding/encoder/__init__.py:203-229 for
slide / patient embeddings):
  tile file:   datasets ``coords`` f32 [N, 2] (micrometres, top-left) and ``feats`` f16 [N, D]; root attributes ``stamp_version``
               (str), ``extractor`` (str), ``unit`` = "um", ``tile_size_um`` (float), ``tile_size_px`` (int), ``code_hash`` (str),
               ``feat_type`` = "tile"
  slide file:  dataset ``feats`` [D]; attributes ``version``, ``encoder``, ``precision``, ``stamp_version``, ``code_hash``,
               ``feat_type`` in {"slide", "patient"}
Files are written under a temporary name and renamed (the reference does the same to avoid half-written files, :344-366).

Readers restate the reference's loaders: `detect_feature_type` (src/stamp/modeling/data.py:424-456), `get_coords` with its three
coordinate conventions -- STAMP v2 (``tile_size`` + ``unit == "um"``), the current one (``tile_size_um``), and the historic
256 um / 224 px one detected from a stride of ~224 (:741-808) -- and the missing-``coords`` bypass (:743-757).

Backend: `h5py` when it can be imported (the reference's dependency); otherwise the same C library h5py wraps, ``libhdf5``,
through ctypes -- strings are written as variable-length UTF-8, Python floats / ints as 64-bit scalars, exactly what h5py does,
so the files interchange with the reference's.  If neither is on the machine (e.g. a bare GPU box), `stamp_amd.h5min` -- a pure-Python
implementation of the subset of the format these files use -- takes over (`backend()`); files of one backend are read by the others.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import tempfile
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import h5min

try:                                    # the reference's own dependency, when present
    import h5py as _h5py
except Exception:                       # noqa: BLE001
    _h5py = None

STAMP_FORMAT = "2.5.0"          # the STAMP release whose file format this package reads and writes (pyproject.toml:3); = encoder.STAMP_FORMAT_VERSION
_HID = C.c_int64
_LIB = None


def _find_libhdf5():
    names = [os.environ.get("AMDSTAMP_LIBHDF5", ""), ctypes.util.find_library("hdf5") or "", "libhdf5.so", "libhdf5_serial.so"]
    import glob
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*"):
        names += sorted(glob.glob(pat), key=len)
    for n in names:
        if not n:
            continue
        try:
            return C.CDLL(n)
        except OSError:
            continue
    return None


def backend() -> str:
    """Which implementation reads / writes the files: "h5py" when it imports (the reference's dependency), else "c" = libhdf5 through
    ctypes when the shared library is on the machine, else "min" = the pure-Python subset in `stamp_amd.h5min` (enough for STAMP's feature
    files).  AMDSTAMP_H5_BACKEND=h5py|c|min forces one (tests)."""
    forced = os.environ.get("AMDSTAMP_H5_BACKEND", "")
    if forced:
        if forced not in ("h5py", "c", "min") or (forced == "h5py" and _h5py is None):
            raise RuntimeError(f"AMDSTAMP_H5_BACKEND={forced!r} is not available")
        return forced
    if _h5py is not None:
        return "h5py"
    global _LIB
    if _LIB is None and _find_libhdf5() is None:
        return "min"
    return "c"


def _lib():
    """libhdf5 with the handful of prototypes used here (HDF5 >= 1.10: hid_t is 64-bit)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    lib = _find_libhdf5()
    if lib is None:
        raise RuntimeError("neither h5py nor libhdf5 is available: cannot read or write STAMP feature files")
    sz, i, u, vp, cp = C.c_size_t, C.c_int, C.c_uint, C.c_void_p, C.c_char_p
    hs = C.c_uint64
    P = {
        "H5open": (i, []), "H5get_libversion": (i, [C.POINTER(u)] * 3), "H5free_memory": (i, [vp]),
        "H5Fcreate": (_HID, [cp, u, _HID, _HID]), "H5Fopen": (_HID, [cp, u, _HID]), "H5Fclose": (i, [_HID]),
        "H5Screate": (_HID, [i]), "H5Screate_simple": (_HID, [i, C.POINTER(hs), C.POINTER(hs)]), "H5Sclose": (i, [_HID]),
        "H5Sget_simple_extent_ndims": (i, [_HID]), "H5Sget_simple_extent_dims": (i, [_HID, C.POINTER(hs), C.POINTER(hs)]),
        "H5Tcopy": (_HID, [_HID]), "H5Tclose": (i, [_HID]), "H5Tset_size": (i, [_HID, sz]), "H5Tget_size": (sz, [_HID]),
        "H5Tset_fields": (i, [_HID, sz, sz, sz, sz, sz]), "H5Tset_ebias": (i, [_HID, sz]), "H5Tset_cset": (i, [_HID, i]), "H5Tset_strpad": (i, [_HID, i]),
        "H5Tget_class": (i, [_HID]), "H5Tis_variable_str": (i, [_HID]), "H5Tget_sign": (i, [_HID]),
        "H5Dcreate2": (_HID, [_HID, cp, _HID, _HID, _HID, _HID, _HID]), "H5Dopen2": (_HID, [_HID, cp, _HID]), "H5Dclose": (i, [_HID]),
        "H5Dwrite": (i, [_HID, _HID, _HID, _HID, _HID, vp]), "H5Dread": (i, [_HID, _HID, _HID, _HID, _HID, vp]),
        "H5Dget_space": (_HID, [_HID]), "H5Dget_type": (_HID, [_HID]),
        "H5Acreate2": (_HID, [_HID, cp, _HID, _HID, _HID, _HID]), "H5Awrite": (i, [_HID, _HID, vp]), "H5Aclose": (i, [_HID]),
        "H5Aopen_by_idx": (_HID, [_HID, cp, i, i, hs, _HID, _HID]), "H5Aget_name": (C.c_ssize_t, [_HID, sz, cp]),
        "H5Aget_type": (_HID, [_HID]), "H5Aget_space": (_HID, [_HID]), "H5Aread": (i, [_HID, _HID, vp]), "H5Aget_num_attrs": (i, [_HID]),
        "H5Lexists": (i, [_HID, cp, _HID]), "H5Gopen2": (_HID, [_HID, cp, _HID]), "H5Gclose": (i, [_HID]), "H5Eset_auto2": (i, [_HID, vp, vp]),
    }
    for name, (res, args) in P.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise RuntimeError("H5open failed")
    lib.H5Eset_auto2(0, None, None)            # errors are reported through return codes -> Python exceptions, not stderr dumps
    _LIB = lib
    return lib


def _g(name: str) -> int:
    return _HID.in_dll(_lib(), name).value


def _chk(rc, what: str):
    if rc < 0:
        raise RuntimeError(f"libhdf5: {what} failed")
    return rc


_F_TRUNC, _F_RDONLY, _S_SCALAR, _VARIABLE, _UTF8 = 2, 0, 0, C.c_size_t(-1).value, 1
_CLS_INT, _CLS_FLOAT, _CLS_STRING = 0, 1, 3


def _f16_type(lib) -> int:
    t = _chk(lib.H5Tcopy(_g("H5T_IEEE_F32LE_g")), "H5Tcopy")
    _chk(lib.H5Tset_fields(t, 15, 10, 5, 0, 10), "H5Tset_fields")
    _chk(lib.H5Tset_size(t, 2), "H5Tset_size")
    _chk(lib.H5Tset_ebias(t, 15), "H5Tset_ebias")
    return t


def _np_type(lib, arr: np.ndarray):
    """(file/memory type id, owned?) for a little-endian numpy array."""
    k = arr.dtype
    if k == np.float16:
        return _f16_type(lib), True
    table = {np.dtype("<f4"): "H5T_IEEE_F32LE_g", np.dtype("<f8"): "H5T_IEEE_F64LE_g", np.dtype("<i8"): "H5T_STD_I64LE_g", np.dtype("<i4"): "H5T_STD_I32LE_g",
             np.dtype("u1"): "H5T_STD_U8LE_g"}
    if k not in table:
        raise TypeError(f"unsupported dtype {k}")
    return _g(table[k]), False


class _CWriter:
    def __init__(self, path: str):
        self.lib = _lib()
        self.f = _chk(self.lib.H5Fcreate(path.encode(), _F_TRUNC, 0, 0), f"H5Fcreate({path})")

    def dataset(self, name: str, arr: np.ndarray) -> None:
        lib = self.lib
        arr = np.ascontiguousarray(arr)
        dims = (C.c_uint64 * max(arr.ndim, 1))(*arr.shape)
        sp = _chk(lib.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else lib.H5Screate(_S_SCALAR), "H5Screate_simple")
        t, own = _np_type(lib, arr)
        d = _chk(lib.H5Dcreate2(self.f, name.encode(), t, sp, 0, 0, 0), f"H5Dcreate2({name})")
        if arr.size:
            _chk(lib.H5Dwrite(d, t, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), f"H5Dwrite({name})")
        lib.H5Dclose(d)
        lib.H5Sclose(sp)
        if own:
            lib.H5Tclose(t)

    def attr(self, name: str, value) -> None:
        lib = self.lib
        sp = _chk(lib.H5Screate(_S_SCALAR), "H5Screate")
        if isinstance(value, str):               # h5py: variable-length UTF-8 string
            t = _chk(lib.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
            _chk(lib.H5Tset_size(t, _VARIABLE), "H5Tset_size")
            _chk(lib.H5Tset_cset(t, _UTF8), "H5Tset_cset")
            buf = C.c_char_p(value.encode("utf-8"))
            a = _chk(lib.H5Acreate2(self.f, name.encode(), t, sp, 0, 0), f"H5Acreate2({name})")
            _chk(lib.H5Awrite(a, t, C.byref(buf)), f"H5Awrite({name})")
            lib.H5Tclose(t)
        elif isinstance(value, (bool, int, np.integer)):
            v = C.c_int64(int(value))
            t = _g("H5T_STD_I64LE_g")
            a = _chk(lib.H5Acreate2(self.f, name.encode(), t, sp, 0, 0), f"H5Acreate2({name})")
            _chk(lib.H5Awrite(a, _g("H5T_NATIVE_INT64_g"), C.byref(v)), f"H5Awrite({name})")
        elif isinstance(value, (float, np.floating)):
            v = C.c_double(float(value))
            t = _g("H5T_IEEE_F64LE_g")
            a = _chk(lib.H5Acreate2(self.f, name.encode(), t, sp, 0, 0), f"H5Acreate2({name})")
            _chk(lib.H5Awrite(a, _g("H5T_NATIVE_DOUBLE_g"), C.byref(v)), f"H5Awrite({name})")
        else:
            raise TypeError(f"attribute {name}: unsupported type {type(value)}")
        lib.H5Aclose(a)
        lib.H5Sclose(sp)

    def close(self) -> None:
        _chk(self.lib.H5Fclose(self.f), "H5Fclose")


def _c_read(path: str) -> tuple[dict[str, np.ndarray], dict]:
    lib = _lib()
    f = _chk(lib.H5Fopen(str(path).encode(), _F_RDONLY, 0), f"H5Fopen({path})")
    try:
        dsets = {}
        for name in ("feats", "coords", "patch_embeddings"):
            if lib.H5Lexists(f, name.encode(), 0) <= 0:
                continue
            d = _chk(lib.H5Dopen2(f, name.encode(), 0), f"H5Dopen2({name})")
            sp, t = lib.H5Dget_space(d), lib.H5Dget_type(d)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (C.c_uint64 * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            cls, size = lib.H5Tget_class(t), lib.H5Tget_size(t)
            if cls == _CLS_FLOAT:
                dt = {2: np.float16, 4: np.float32, 8: np.float64}[size]
            elif cls == _CLS_INT:
                dt = np.dtype(("i" if lib.H5Tget_sign(t) else "u") + str(size))
            else:
                raise TypeError(f"{path}: dataset {name} has unsupported class {cls}")
            arr = np.empty(shape, dtype=dt)
            if arr.size:
                _chk(lib.H5Dread(d, t, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), f"H5Dread({name})")       # memory type = file type: raw little-endian copy
            dsets[name] = arr
            lib.H5Tclose(t); lib.H5Sclose(sp); lib.H5Dclose(d)
        attrs = {}
        root = _chk(lib.H5Gopen2(f, b"/", 0), "H5Gopen2(/)")
        for idx in range(_chk(lib.H5Aget_num_attrs(root), "H5Aget_num_attrs")):
            a = _chk(lib.H5Aopen_by_idx(root, b".", 0, 0, idx, 0, 0), "H5Aopen_by_idx")
            n = lib.H5Aget_name(a, 0, None)
            nb = C.create_string_buffer(n + 1)
            lib.H5Aget_name(a, n + 1, nb)
            t = lib.H5Aget_type(a)
            cls = lib.H5Tget_class(t)
            if cls == _CLS_STRING:
                if lib.H5Tis_variable_str(t) > 0:
                    p = C.c_void_p()
                    _chk(lib.H5Aread(a, t, C.byref(p)), "H5Aread")
                    val = C.string_at(p).decode("utf-8") if p.value else ""
                    if p.value:
                        lib.H5free_memory(p)
                else:
                    buf = C.create_string_buffer(lib.H5Tget_size(t) + 1)
                    _chk(lib.H5Aread(a, t, buf), "H5Aread")
                    val = buf.value.decode("utf-8", "replace")
            elif cls == _CLS_INT:
                v = C.c_int64()
                _chk(lib.H5Aread(a, _g("H5T_NATIVE_INT64_g"), C.byref(v)), "H5Aread")
                val = int(v.value)
            elif cls == _CLS_FLOAT:
                v = C.c_double()
                _chk(lib.H5Aread(a, _g("H5T_NATIVE_DOUBLE_g"), C.byref(v)), "H5Aread")
                val = float(v.value)
            else:
                val = None
            attrs[nb.value.decode()] = val
            lib.H5Tclose(t); lib.H5Aclose(a)
        lib.H5Gclose(root)
        return dsets, attrs
    finally:
        lib.H5Fclose(f)


# ---- public API ------------------------------------------------------------------------------------------------------------------
def _write(path: Path, datasets: dict[str, np.ndarray], attrs: dict) -> None:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    fd, tmp = tempfile.mkstemp(dir=path.parent)          # intermediate name: no half-written files (preprocessing/__init__.py:344-366)
    os.close(fd)
    try:
        be = backend()
        if be == "h5py":
            with _h5py.File(tmp, "w") as f:
                for k, v in datasets.items():
                    f[k] = v
                for k, v in attrs.items():
                    f.attrs[k] = v
        elif be == "min":
            h5min.write(tmp, {k: np.asarray(v) for k, v in datasets.items()}, attrs)
        else:
            w = _CWriter(tmp)
            try:
                for k, v in datasets.items():
                    w.dataset(k, v)
                for k, v in attrs.items():
                    w.attr(k, v)
            finally:
                w.close()
        os.replace(tmp, path)
    except Exception:
        Path(tmp).unlink(missing_ok=True)
        raise


def _np(x) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _pep440(stamp_version) -> str:
    """The reference parses `stamp_version` with `packaging.version.Version` (modeling/data.py:793-795): refuse anything it would choke on."""
    if not _is_pep440(str(stamp_version)):
        raise ValueError(f"stamp_version must be a PEP 440 version string (STAMP parses it with packaging.Version), got {stamp_version!r}")
    return str(stamp_version)


def _version_newer(a: str, b: str) -> bool:
    """a > b as the reference compares them (packaging's Version); without `packaging`: by release tuple, a pre- / dev-release below its release."""
    try:
        from packaging.version import Version
        return Version(a) > Version(b)
    except ImportError:
        def key(v):
            m = _is_pep440(v) and _PEP440_RE.match(v)
            if not m:
                raise ValueError(f"not a PEP 440 version: {v!r}")
            rel = [int(x) for x in m.group("release").split(".")]
            while len(rel) > 1 and rel[-1] == 0:
                rel.pop()
            return (int(m.group("epoch") or 0), tuple(rel), 0 if (m.group("dev") or m.group("pre")) else (2 if m.group("post") else 1))
        return key(a) > key(b)


# PEP 440, appendix B ("Parsing version strings with regular expressions"): what packaging.version.Version accepts.  Used when `packaging` is not
# importable (a bare GPU box: h5min needs nothing but numpy, and so must the attribute check in front of it).
_PEP440_RE = None


def _is_pep440(v: str) -> bool:
    try:
        from packaging.version import InvalidVersion, Version
    except ImportError:
        global _PEP440_RE
        if _PEP440_RE is None:
            import re
            _PEP440_RE = re.compile(
                r"^\s*v?(?:(?:(?P<epoch>[0-9]+)!)?(?P<release>[0-9]+(?:\.[0-9]+)*)(?P<pre>[-_\.]?(?P<pre_l>alpha|a|beta|b|preview|pre|c|rc)[-_\.]?(?P<pre_n>[0-9]+)?)?"
                r"(?P<post>(?:-(?P<post_n1>[0-9]+))|(?:[-_\.]?(?P<post_l>post|rev|r)[-_\.]?(?P<post_n2>[0-9]+)?))?"
                r"(?P<dev>[-_\.]?(?P<dev_l>dev)[-_\.]?(?P<dev_n>[0-9]+)?)?)(?:\+(?P<local>[a-z0-9]+(?:[-_\.][a-z0-9]+)*))?\s*$", re.IGNORECASE)
        return _PEP440_RE.match(v) is not None
    try:
        Version(v)
        return True
    except InvalidVersion:
        return False


def _build_attrs(amdstamp_version) -> dict:
    return {} if amdstamp_version is None else {"amdstamp_version": str(amdstamp_version)}


def write_tile_features(path, feats, coords_um, *, extractor: str, tile_size_um: float, tile_size_px: int, code_hash: str, stamp_version: str,
                        amdstamp_version: str | None = None) -> None:
    """feats [N, D] (stored as given: the extraction loop hands fp16, preprocessing/__init__.py:325), coords [N, 2] micrometres."""
    feats, coords = _np(feats), _np(coords_um).astype(np.float32)
    if feats.ndim != 2 or coords.shape != (feats.shape[0], 2):
        raise ValueError(f"expected feats [N, D] and coords [N, 2], got {feats.shape} and {coords.shape}")
    _write(Path(path), {"coords": coords, "feats": feats},
           {"stamp_version": _pep440(stamp_version), "extractor": str(extractor), "unit": "um", "tile_size_um": float(tile_size_um),
            "tile_size_px": int(tile_size_px), "code_hash": str(code_hash), "feat_type": "tile", **_build_attrs(amdstamp_version)})


def write_slide_features(path, feats, *, encoder: str, precision: str, code_hash: str, stamp_version: str, feat_type: str = "slide",
                         amdstamp_version: str | None = None) -> None:
    if feat_type not in ("slide", "patient"):
        raise ValueError("feat_type must be 'slide' or 'patient'")
    _write(Path(path), {"feats": _np(feats)}, {"version": _pep440(stamp_version), "encoder": str(encoder), "precision": str(precision),
                                               "stamp_version": _pep440(stamp_version), "code_hash": str(code_hash), "feat_type": feat_type,
                                               **_build_attrs(amdstamp_version)})


def read_file(path) -> tuple[dict[str, np.ndarray], dict]:
    """(datasets among feats / coords / patch_embeddings, root attributes)."""
    be = backend()
    if be == "h5py":
        with _h5py.File(path, "r") as f:
            return read_open_h5py(f)
    if be == "min":
        return h5min.read(path, want=("feats", "coords", "patch_embeddings"))
    return _c_read(str(path))


def read_open_h5py(f) -> tuple[dict[str, np.ndarray], dict]:
    """`read_file` on an h5py handle that is already open (stamp_amd.bags keeps handles open between reads, as the reference does): only the
    three datasets STAMP reads, attributes normalised (bytes decoded, numpy scalars as Python scalars)."""
    d = {k: f[k][()] for k in ("feats", "coords", "patch_embeddings") if k in f}
    a = {k: (v.decode() if isinstance(v, bytes) else (v.item() if hasattr(v, "item") and getattr(v, "size", 1) == 1 else v)) for k, v in f.attrs.items()}
    return d, a


def feature_type(attrs: dict) -> str:
    """`detect_feature_type`'s per-file rule (data.py:438-446)."""
    ft, enc = attrs.get("feat_type"), attrs.get("encoder")
    return str(ft) if (ft is not None or enc is not None) else "tile"


@dataclass
class CoordsInfo:
    coords_um: np.ndarray
    tile_size_um: float
    tile_size_px: int | None = None

    @property
    def mpp(self) -> float:
        if not self.tile_size_px:
            raise RuntimeError("tile size in pixels is not available. Please reextract them using `stamp preprocess`.")
        return self.tile_size_um / self.tile_size_px


def _stride(coords: np.ndarray) -> float:
    """reference `get_stride` (data.py:1150-1161): the smallest step between sorted unique coordinates along either axis."""
    best = np.inf
    for ax in range(coords.shape[1]):
        u = np.unique(coords[:, ax].astype(np.float32))
        if u.size > 1:
            best = min(best, float(np.diff(u).min()))
    return best


def get_coords(datasets: dict[str, np.ndarray], attrs: dict) -> CoordsInfo:
    """The reference's `get_coords` (data.py:741-808) on an opened file's contents; pinned by tests/golden/get_coords.json (the reference's own function)."""
    if "coords" not in datasets:                               # multiplex bypass (:743-757)
        n = datasets["patch_embeddings"].shape[0]
        return CoordsInfo(np.stack([np.arange(n), np.zeros(n)], axis=1).astype(np.float32), 0.0, 0)
    coords = datasets["coords"]
    tile_um = tile_px = coords_um = None
    if attrs.get("tile_size") and attrs.get("unit") == "um":   # STAMP v2 format
        tile_um, coords_um = float(attrs["tile_size"]), coords
    elif attrs.get("tile_size_um"):                            # newer format
        tile_um, coords_um = float(attrs["tile_size_um"]), coords
    elif round(attrs.get("tile_size", _stride(coords))) == 224:   # historic format: coordinates in units of 256 um / 224 px
        tile_um, tile_px, coords_um = 256.0, 224, coords / 224 * 256
    if attrs.get("stamp_version"):                             # a file from a newer STAMP than the format this package speaks is refused (:793-799)
        if _version_newer(str(attrs["stamp_version"]), STAMP_FORMAT):
            raise RuntimeError(f"features were extracted with a newer version of stamp, please update your stamp to at least version {attrs['stamp_version']}.")
    if not tile_px and "tile_size_px" in attrs:
        tile_px = int(attrs["tile_size_px"])
    if not tile_um or coords_um is None:
        raise RuntimeError("unable to infer coordinates from feature file. Please reextract them using `stamp preprocess`.")
    return CoordsInfo(np.asarray(coords_um), tile_um, tile_px)


def read_tile_features(path):
    """-> (feats [N, D] as stored, CoordsInfo, attrs)."""
    d, a = read_file(path)
    feats = d["feats"] if "feats" in d else d["patch_embeddings"]
    return feats, get_coords(d, a), a
