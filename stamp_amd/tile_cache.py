"""STAMP's tile-cache zip format (host I/O): reading caches an unmodified STAMP wrote, and writing ones it can read.

Reference src/stamp/preprocessing/tiling.py: `tiles_with_cache` :68-168 (a zip per slide named `<slide stem>.<sha256 of the tiler
parameters>.zip`, written under a temporary name and renamed; entry `tiler_params.json` first, then one image per tile named
``tile_({float(x_um)}, {float(y_um)}).{ext}`` with ext in {"jpg", "png"}; PNGs saved without ICC profile :143-151),
`_tiles_from_cache_file` :380-406 (entries matched by the regex ``tile_\\((\\d+\\.\\d+), (\\d+\\.\\d+)\\).<ext>``; `tile_ext` defaults to
"jpg" for caches written before the field existed), `_TilerParams` :356-378.

Image decoding / encoding is PIL on the host, like the reference.  Decoded tiles come back as one u8 [N, H, W, 3] array ready for
`stamp_amd.extractor.extract_tiles` (pinned H2D -> HIP tile encoder).
"""
from __future__ import annotations

import hashlib
import io
import json
import os
import re
import tempfile
from pathlib import Path
from zipfile import ZipFile

import numpy as np

EXTENSION_TO_FORMAT = {"png": "png", "jpg": "jpeg"}            # reference src/stamp/types.py:17-21


def tiler_params(slide_path, *, tile_size_um: float, tile_size_px: int, max_supertile_size_slide_px: int, brightness_cutoff, code_sha256: str,
                 tile_ext: str) -> dict:
    """`_TilerParams` in the reference's key order (:96-104).  `code_sha256` identifies the tiling code; STAMP uses the hash of its own
    tiling.py, so a cache written by this package is never mistaken for one written by STAMP (and vice versa) unless the caller passes
    STAMP's value on purpose."""
    return {"slide_path": str(slide_path), "tile_size_um": tile_size_um, "tile_size_px": tile_size_px,
            "max_supertile_size_slide_px": max_supertile_size_slide_px, "brightness_cutoff": brightness_cutoff, "code_sha256": code_sha256,
            "tile_ext": tile_ext}


def cache_file_path(cache_dir, slide_path, params: dict) -> Path:
    """:105-110: `<cache_dir>/<slide name with its suffix replaced by .<sha256(json.dumps(params, sort_keys=True))>.zip>`."""
    h = hashlib.sha256(json.dumps(params, sort_keys=True).encode()).hexdigest()
    return Path(cache_dir) / Path(slide_path).with_suffix(f".{h}.zip").name


def read_tile_cache(path):
    """-> (tiles u8 [N, H, W, 3], coords_um float64 [N, 2], tiler params dict), entries in the zip's own order (:380-406)."""
    from PIL import Image

    tiles, coords = [], []
    with ZipFile(path, "r") as z:
        params = json.loads(z.read("tiler_params.json").decode())
        ext = params.get("tile_ext", "jpg")                        # "jpg" for backwards compatibility (:386-387)
        for name in z.namelist():
            m = re.match(rf"tile_\((\d+\.\d+), (\d+\.\d+)\).{ext}", name)
            if m is None:
                continue
            with z.open(name, "r") as fp:
                img = Image.open(fp)
                img.load()
            tiles.append(np.asarray(img.convert("RGB"), dtype=np.uint8))
            coords.append((float(m.group(1)), float(m.group(2))))
    if not tiles:
        return np.zeros((0, 0, 0, 3), np.uint8), np.zeros((0, 2), np.float64), params
    return np.stack(tiles), np.array(coords, dtype=np.float64), params


def write_tile_cache(path, tiles_u8: np.ndarray, coords_um: np.ndarray, params: dict) -> None:
    """Writes the zip atomically (temporary file + rename, :118-168).  tiles u8 [N, H, W, 3]; coords [N, 2] micrometres."""
    from PIL import Image

    ext = params.get("tile_ext", "jpg")
    if ext not in EXTENSION_TO_FORMAT:
        raise ValueError(f"tile_ext must be one of {list(EXTENSION_TO_FORMAT)}, got {ext!r}")
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    fd, tmp = tempfile.mkstemp(dir=path.parent)
    os.close(fd)
    try:
        with ZipFile(tmp, "w") as z:
            with z.open("tiler_params.json", "w") as fp:
                fp.write(json.dumps(params).encode())
            for tile, (x, y) in zip(tiles_u8, coords_um):
                buf = io.BytesIO()
                Image.fromarray(np.ascontiguousarray(tile), "RGB").save(buf, format=EXTENSION_TO_FORMAT[ext], **(dict(icc_profile=None) if ext == "png" else {}))
                with z.open(f"tile_({float(x)}, {float(y)}).{ext}", "w") as fp:
                    fp.write(buf.getvalue())
        os.replace(tmp, path)
    except Exception:
        Path(tmp).unlink(missing_ok=True)
        raise
